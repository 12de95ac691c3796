#!/usr/bin/env python3
"""How long the host needs to ISSUE one bench sweep (no synchronisation inside), against the sweep's GPU time.

    python tools/host_time.py [sweeps]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd.Demix import dNMF as M  # noqa: E402
from dnmf_amd.WUtils import Simulator  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    size, K, T, bs = 512, 100, 4000, 4
    sz = [size, size, 1]
    torch.manual_seed(0)
    np.random.seed(0)
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=positions[:, :, 0].contiguous())
    dn.verbose = False
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-5 * (50.0 / size) ** 2)
    train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(1234))
    test = M.ResidentLoader(frames, sz, bs, shuffle=False)

    def step():
        dn.update_motion(train, opt, gamma=1, epochs=1)
        dn.update_footprints(test, bs, sz, gamma_c=0, gamma_a=1.0, iter_c=50, return_dense=False)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    issue, total = [], []
    for _ in range(n):
        t0 = time.perf_counter()
        step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append(1e3 * (t1 - t0))
        total.append(1e3 * (t2 - t0))
    issue, total = np.array(issue), np.array(total)
    print(f"host issue time per sweep: median {np.median(issue):.2f} ms, max {issue.max():.2f}; "
          f"sweep incl. wait: median {np.median(total):.2f} ms, max {total.max():.2f}")
    print("issue:", " ".join(f"{v:.1f}" for v in issue))
    print("total:", " ".join(f"{v:.1f}" for v in total))


if __name__ == "__main__":
    main()
