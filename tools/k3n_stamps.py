#!/usr/bin/env python3
"""Where a wave of K3n spends its cycles (diagnostic build with s_memtime stamps):

    python -m dnmf_amd.build --out build/variants/k3n_stamps.so -DDNMF_K3N_STAMPS=1
    DNMF_LIB=build/variants/k3n_stamps.so python tools/k3n_stamps.py [frames]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops  # noqa: E402
from dnmf_amd.Demix import dNMF as M  # noqa: E402
from dnmf_amd.WUtils import Simulator  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    size, K = 512, 100
    sz = [size, size, 1]
    torch.manual_seed(0)
    np.random.seed(0)
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=positions[:, :, 0].contiguous())
    with torch.no_grad():
        dn.fp.beta += 1e-6 * torch.randn_like(dn.fp.beta)
    ly = dn.fp.packed_lists()
    ops.TIMING = {}
    ws = None
    for rep in range(4):
        if ops.LISTS_COUNTERS is not None:
            ops.LISTS_COUNTERS.zero_()
        _, _, ws = ops.warp_gram_rhs_lists(ly, K, sz, dn.fp.beta.detach(), None, frames, workspace=ws, finish=False)
        torch.cuda.synchronize()
    c = ops.LISTS_COUNTERS.tolist()
    (a, b) = ops.TIMING["warp_gram_rhs_lists"][-1]
    ms = a.elapsed_time(b)
    tot = sum(c[2:9])
    names = ["tile bookkeeping", "reductions of finished runs", "coordinates / weights / frame loads issued",
             "regions requested, arrived, stored", "taps from LDS + per-lane sums", "rest (prologue, table write-back)",
             "long-list tiles after their coordinates"]
    print(f"launch {ms:.3f} ms; evals/frame {c[0] / T:.0f}; wave-cycles total {tot:.3e}")
    print(f"non-empty tiles per frame {c[9] / T:.0f} of 1024; long lists {c[10] / T:.1f}; flushed runs {c[11] / T:.0f}")
    for n, v in zip(names, c[2:9]):
        print(f"  {n:45s} {100.0 * v / max(tot, 1):5.1f} %   {v / (T * 1024):8.0f} cycles per tile")


if __name__ == "__main__":
    main()
