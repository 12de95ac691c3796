import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dnmf_amd.Demix import dNMF as M
torch.manual_seed(3)
rng = np.random.RandomState(3)
T, bs = 16, 4
sz, K = [24, 20, 2], 6
pos = rng.rand(K, 3) * np.array(sz)
frames = torch.rand(T, sz[0] * sz[1] * sz[2], device="cuda")
C0 = torch.rand(K, T)
dns, opts, loaders = [], [], []
for fused in (False, True):
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
    dn.verbose, dn.fused_motion = False, fused
    dn.C = C0.to("cuda")
    dns.append(dn)
    opts.append(torch.optim.Adam([dn.fp.beta], lr=1e-3))
    loaders.append(M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(5)))
for ep in range(5):
    for dn, opt, ld in zip(dns, opts, loaders):
        dn.update_motion(ld, opt, gamma=1, epochs=1)
    b0, b1 = dns[0].fp.beta.detach(), dns[1].fp.beta.detach()
    d = (b0 - b1).abs()
    idx = torch.nonzero(d > 1e-6)
    print("epoch", ep, "max diff", float(d.max()), "n>1e-6:", idx.shape[0], "frames:", sorted(set(idx[:, 2].tolist())),
          "rows:", sorted(set(idx[:, 0].tolist())))
    g0, g1 = dns[0].fp.beta.grad, dns[1].fp.beta.grad
    # last batch's grad (stepwise) vs same frames in the fused full grad
    nzf = torch.nonzero(g0.abs().sum((0, 1)) > 0).flatten()
    print("   grad rel diff on last batch frames", float((g0[:, :, nzf] - g1[:, :, nzf]).abs().max() / g0.abs().max()))
