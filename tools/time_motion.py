#!/usr/bin/env python3
"""The motion gradient of all frames at the bench geometry: reconstruction images of all frames through HBM
(dnmf_recon_image_lists + dnmf_warp_recon_grad) against dnmf_motion_grad_lists with pieces of various sizes.

    python tools/time_motion.py [frames] [chunk ...]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops  # noqa: E402
from dnmf_amd.Demix import dNMF as M  # noqa: E402


def timed(fn, reps=6):
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return ms


def main():
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    chunks = [int(c) for c in sys.argv[2:]] or [32, 64, 96, 128, 192, 256, 512]
    size, K = 512, 100
    sz = [size, size, 1]
    torch.manual_seed(0)
    pos = torch.rand(K, 3) * torch.tensor([float(size), float(size), 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    with torch.no_grad():
        scale = torch.tensor([1.0, 1e-3, 1e-3, 1e-3, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6], device="cuda")
        fp.beta += scale[:, None, None] * torch.randn_like(fp.beta)
    C = torch.rand(K, T, device="cuda")
    frames = torch.rand(T, fp.P, device="cuda")
    times = torch.arange(T, dtype=torch.int32, device="cuda")
    ly = fp.packed_lists()
    beta = fp.beta.detach()
    S = torch.empty((T, ops.halo_voxels(sz)), device="cuda")
    state = {"ws": None}

    def separate():
        g = torch.zeros_like(beta)
        ops.recon_image_lists(ly, K, sz, C, times, out=S)
        out = ops.warp_recon_grad(S, times, frames, times, sz, beta, times, grad=g, want_loss=False, want_reg=False,
                                  workspace=state["ws"], norm_frames=4)
        state["ws"] = out["workspace"]
        return g

    ref = separate()
    ms = timed(separate)
    print("all frames through HBM: ", " ".join("%.3f" % m for m in ms), "ms")
    del S
    for c in chunks:
        st = {"ws": None}

        def run():
            g = torch.zeros_like(beta)
            out = ops.motion_grad_lists(ly, K, sz, C, frames, times, beta, times, g, 4, c, workspace=st["ws"])
            st["ws"] = out["workspace"]
            return g

        g = run()
        same = torch.equal(g, ref)
        ms = timed(run)
        print("pieces of %4d frames:   " % c, " ".join("%.3f" % m for m in ms), "ms   bit-identical gradient:", same)


if __name__ == "__main__":
    main()
