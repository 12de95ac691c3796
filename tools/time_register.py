#!/usr/bin/env python3
"""Time K8 (dnmf_register_patches, dnmf_rigid_correct) on a synthetic video: python tools/time_register.py [size] [Z] [T] [stride] [overlap]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops  # noqa: E402


def main():
    a = [int(v) for v in sys.argv[1:]]
    size, Z, T, stride, overlap = (a + [512, 2, 256, 96, 32][len(a):])[:5]
    sz = [size, size, Z]
    torch.manual_seed(0)
    frames = torch.rand(T, size * size * Z, device="cuda")
    tmpl = frames.mean(0)
    st, ov, ms = (stride, stride, 1), (overlap, overlap, Z - 1), (6, 6, 1 if Z > 1 else 0)
    dims, starts = ops.patch_grid(sz, st, ov)
    for i in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rigid, patch = ops.register_patches(frames, tmpl, sz, st, ov, ms, 3, 10, 0.0)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{sz} T={T} patches {tuple(dims)} of {tuple(s + o for s, o in zip(st, ov))}: {1e3 * dt:.1f} ms = {T / dt:.0f} frames/s", flush=True)
    for i in range(3):   # the rigid pass behind template=None: shifts, every frame moved through its spectrum, sums for the template
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ops.rigid_correct(frames, tmpl, sz, ms, 10, 0.0, True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{sz} T={T} rigid correction: {1e3 * dt:.1f} ms = {T / dt:.0f} frames/s", flush=True)


if __name__ == "__main__":
    main()
