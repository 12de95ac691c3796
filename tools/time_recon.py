#!/usr/bin/env python3
"""Time the two reconstruction-image kernels at the bench geometry."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from dnmf_amd import ops
    from dnmf_amd.Demix import dNMF as M
    torch.manual_seed(0)
    size, K, T = 512, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    sz = [size, size, 1]
    pos = torch.rand(K, 3) * torch.tensor([float(size), float(size), 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    C = torch.rand(K, T, device="cuda")
    times = torch.arange(T, dtype=torch.int32, device="cuda")
    lds = ops.halo_voxels(sz)
    S = torch.empty((T, lds), device="cuda")
    ly = fp.packed_lists()
    for name, fn in (("mfma", lambda: ops.recon_image(fp.packed_footprints(), K, sz, C, times, out=S)),
                     ("lists", lambda: ops.recon_image_lists(ly, K, sz, C, times, out=S))):
        ms = []
        for _ in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        print(name, ["%.2f" % m for m in ms], "ms ->", "%.2f TB/s written" % (4.0 * lds * T / (min(ms) * 1e-3) / 1e12))


if __name__ == "__main__":
    main()
