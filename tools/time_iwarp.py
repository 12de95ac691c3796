#!/usr/bin/env python3
"""Time K7 (dnmf_image_iwarp) alone: python tools/time_iwarp.py [frames] [size] [depth] [warp amplitude in px]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops  # noqa: E402


def main():
    a = [float(v) for v in sys.argv[1:]]
    T, size, Z, amp = int(a[0]) if a else 2000, int(a[1]) if len(a) > 1 else 512, int(a[2]) if len(a) > 2 else 1, a[3] if len(a) > 3 else 0.0
    sz = [size, size, Z]
    P = size * size * Z
    torch.manual_seed(0)
    frames = torch.rand(T, P, device="cuda")
    beta = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].repeat(1, 1, T).cuda()
    scale = torch.tensor([1.0, 1.0 / size, 1.0 / size, 0, 1.0 / size ** 2, 1.0 / size ** 2, 0, 1.0 / size ** 2, 0, 0], device="cuda")
    beta += (amp + 0.01) * torch.randn_like(beta) * scale[:, None, None]
    beta[:, 2] = torch.tensor([0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0], device="cuda")[:, None]
    beta = beta.contiguous()
    times = torch.arange(T, dtype=torch.int32, device="cuda")
    out = torch.empty_like(frames)
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    for i in range(5):
        count.zero_()
        a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a_.record()
        ops.image_iwarp(frames, None, sz, beta, times, out=out, count=count)
        b_.record()
        torch.cuda.synchronize()
        print(os.environ.get("DNMF_LIB", "product"), sz, T, f"amp {amp}: {a_.elapsed_time(b_):.3f} ms, exhaustive points {int(count)}", flush=True)


if __name__ == "__main__":
    main()
