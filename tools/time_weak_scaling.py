#!/usr/bin/env python3
"""Per-rank sweep time as the number of ranks grows (weak scaling), rehearsed on ONE GPU: rank 0's share of a
4000*N-frame video with the global mini-batch plan of N ranks (the other ranks' frames are simply absent)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.set_num_threads(8)
from dnmf_amd.Demix import dNMF as M
from dnmf_amd.WUtils import Simulator

size, K, Tn, bs = 512, 100, 4000, 4
sz = [size, size, 1]
torch.manual_seed(0); np.random.seed(0)
frames, positions, _ = Simulator.generate_video_resident(K, Tn, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
frames.clamp_(min=0)
for world in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    dn = M.DeformableNMF(torch.tensor(sz), K, Tn, positions=positions[:, :, 0].contiguous()); dn.verbose = False
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-5)
    train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(1), t0=0, T_total=Tn * world)
    test = M.ResidentLoader(frames, sz, bs)
    def sweep():
        dn.update_motion(train, opt, gamma=1, epochs=1)
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False)
    for _ in range(5):
        sweep()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        sweep()
    torch.cuda.synchronize()
    print(f"world {world}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms per sweep on rank 0 ({Tn * world // bs} optimiser steps per epoch)", flush=True)
