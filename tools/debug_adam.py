import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dnmf_amd import ops
torch.manual_seed(0)
T, bs, E = 16, 4, 5
beta0 = torch.randn(10, 3, T, device="cuda")
p1 = beta0.clone().requires_grad_(True)
opt = torch.optim.Adam([p1], lr=1e-3)
p2 = beta0.clone()
m = torch.zeros_like(p2); v = torch.zeros_like(p2)
step0 = 0
gen = torch.Generator().manual_seed(1)
for ep in range(E):
    perm = torch.randperm(T, generator=gen)
    batches = [perm[s:s + bs] for s in range(0, T, bs)]
    # per-frame gradients (fixed per epoch, independent of beta)
    G = torch.randn(10, 3, T, device="cuda") * torch.logspace(-8, 0, 10, device="cuda")[:, None, None]
    for b in batches:
        opt.zero_grad()
        g = torch.zeros_like(p1)
        g[:, :, b] = G[:, :, b.cuda()]
        p1.grad = g
        opt.step()
    fs = torch.full((T,), -1, dtype=torch.int32)
    for j, b in enumerate(batches):
        fs[b] = j
    n = len(batches)
    ops.adam_epoch(p2, None, m, v, step0, fs, n, 1e-3, (0.9, 0.999), 1e-8, 0)
    ops.adam_epoch(p2, G.contiguous(), m, v, step0, fs, n, 1e-3, (0.9, 0.999), 1e-8, 1)
    step0 += n
    d = (p1.detach() - p2).abs()
    print(ep, "max |dp|", float(d.max()), "disp", float((p1.detach() - beta0).abs().max()),
          "max rel dm", float(((opt.state[p1]['exp_avg'] - m).abs() / (m.abs() + 1e-30)).max()),
          "max rel dv", float(((opt.state[p1]['exp_avg_sq'] - v).abs() / (v.abs() + 1e-30)).max()))
