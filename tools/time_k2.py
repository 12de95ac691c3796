#!/usr/bin/env python3
"""Time K2 (dnmf_warp_recon_grad, the fit step's call) alone at the bench geometry; DNMF_LIB selects a variant build.

    python tools/time_k2.py [frames] [size] [depth]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops  # noqa: E402


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    Z = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    sz = [size, size, Z]
    P = size * size * Z
    torch.manual_seed(0)
    S = torch.rand((T, ops.halo_voxels(sz)), device="cuda")
    frames = torch.rand((T, P), device="cuda")
    beta = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].repeat(1, 1, T).cuda()
    beta += 1e-4 * torch.randn_like(beta) * torch.tensor([100, 1, 1, 1, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2, 1e-2], device="cuda")[:, None, None]
    beta = beta.contiguous()
    times = torch.arange(T, dtype=torch.int32, device="cuda")
    grad = torch.zeros_like(beta)
    ws = None
    ms = []
    for i in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = ops.warp_recon_grad(S, times, frames, times, sz, beta, times, grad=grad, want_loss=False, want_reg=False,
                                  workspace=ws, norm_frames=4)
        b.record()
        ws = out["workspace"]
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    best = min(ms[2:])
    print(os.environ.get("DNMF_LIB", "product"), sz, T, "frames:", " ".join("%.3f" % m for m in ms), "ms; best %.3f ms = %.2f TB/s"
          % (best, 8.0 * P * T / best / 1e9), "grad checksum %.6e" % float(grad.double().abs().sum()))


if __name__ == "__main__":
    main()
