#!/usr/bin/env python3
"""Does the bench workload stay finite, and how much work does K3n do per frame?

    python tools/bench_sanity.py [frames] [sweeps] [lr]

With the demo's lr = 1e-5 on a 512 x 512 volume the fit diverges within one epoch (whole footprints leave the volume,
their traces overflow, beta turns NaN frame by frame) and K3n has ever less to do; bench.py scales the learning rate
with the volume (1e-5 * (50 / size)^2) and reports the same quantities as `fit_sanity`."""
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
    sweeps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    size, K, bs = 512, 100, 4
    lr = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5 * (50.0 / size) ** 2
    sz = [size, size, 1]
    torch.manual_seed(0)
    np.random.seed(0)
    par = {"sigma": [5, 5, .01], "ls": [10, 10, 10]}
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, par)
    frames.clamp_(min=0)
    pos0 = positions[:, :, 0].contiguous()
    print("positions0 range", pos0.min(0).values.tolist(), pos0.max(0).values.tolist())
    torch.manual_seed(1)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos0)
    dn.verbose = False
    ly = dn.fp.packed_lists()
    bb = ly["bbox"].cpu()
    print("boxfrac", ly["boxfrac"], "nslot", ly["nslot"], "bbox extents x", (bb[:, 1] - bb[:, 0] + 1).tolist()[:10])
    opt = torch.optim.Adam([dn.fp.beta], lr=lr)
    print("lr", lr)
    train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(1234))
    test = M.ResidentLoader(frames, sz, bs)
    ops.TIMING = {}
    for s in range(sweeps):
        if ops.LISTS_COUNTERS is not None:
            ops.LISTS_COUNTERS.zero_()
        dn.update_motion(train, opt, gamma=1, epochs=1)
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False)
        torch.cuda.synchronize()
        cnt = ops.LISTS_COUNTERS.tolist() if ops.LISTS_COUNTERS is not None else None
        ms = {k: sum(a.elapsed_time(b) for a, b in v) for k, v in ops.TIMING.items()}
        ops.TIMING = {}
        b = dn.fp.beta.detach()
        fin_b = torch.isfinite(b).all(0).all(0)
        C = dn.C
        print(f"sweep {s}: finite beta columns {int(fin_b.sum())}/{T}, finite C entries {int(torch.isfinite(C).sum())}/{C.numel()}, "
              f"C max {float(C[torch.isfinite(C)].max()):.3e}, evals/frame {cnt[0] / T if cnt else None}, pairs/frame {cnt[1] / T if cnt else None}, "
              f"ms {({k: round(v, 3) for k, v in ms.items()})}, |beta - id| max {float((b[:, :, fin_b] - torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].cuda()).abs().max()):.3e}")


if __name__ == "__main__":
    main()
