#!/usr/bin/env python3
"""Wall time of the pieces of one bench sweep (synchronised), to find host-side overheads."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops
from dnmf_amd.Demix import dNMF as M
from dnmf_amd.WUtils import Simulator

def T(f, n=3):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    return min(ts), r

size, K, Tn, bs = 512, 100, 4000, 4
sz = [size, size, 1]
torch.manual_seed(0); np.random.seed(0)
frames, positions, _ = Simulator.generate_video_resident(K, Tn, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
frames.clamp_(min=0)
dn = M.DeformableNMF(torch.tensor(sz), K, Tn, positions=positions[:, :, 0].contiguous()); dn.verbose = False
opt = torch.optim.Adam([dn.fp.beta], lr=1e-5)
train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(1))
test = M.ResidentLoader(frames, sz, bs)
dn.update_motion(train, opt, gamma=1, epochs=1); dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False)
print("update_motion      %.2f ms" % T(lambda: dn.update_motion(train, opt, gamma=1, epochs=1))[0])
print("update_footprints  %.2f ms" % T(lambda: dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False))[0])
print("  recon cache      %.2f ms" % T(lambda: dn._recon_cache())[0])
S_all = dn._recon_cache()
print("  motion epoch     %.2f ms" % T(lambda: dn._motion_epoch(train, opt, S_all))[0])
print("  epoch_plan       %.2f ms" % T(lambda: train.epoch_plan())[0])
fr, order = dn._gather_frames(test)
print("  gram_rhs         %.2f ms" % T(lambda: dn._gram_rhs(fr, order))[0])
G, r = dn._gram_rhs(fr, order)
print("  C gather         %.2f ms" % T(lambda: dn.C.to('cuda', torch.float32)[:, order.long()].contiguous())[0])
Csel = dn.C[:, order.long()].contiguous()
print("  mu_temporal      %.2f ms" % T(lambda: M._mu_temporal(G, r, Csel, 0, 50))[0])
print("  C scatter        %.2f ms" % T(lambda: dn.C.clone().index_copy_(1, order.long(), Csel))[0])
def sweep():
    dn.update_motion(train, opt, gamma=1, epochs=1)
    dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False)
for label, timing in (("plain", None), ("with ops.TIMING", {})):
    ops.TIMING = timing
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        sweep()
    torch.cuda.synchronize()
    print("5 sweeps %-16s %.2f ms each" % (label, 1e3 * (time.perf_counter() - t0) / 5))
ops.TIMING = None
# per-sweep wall times: a host stall (CPU quota throttling, allocator re-mapping) shows as an outlier here
ts = []
for _ in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter(); sweep(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print("12 sweeps, each synchronised:", ["%.1f" % t for t in ts])
