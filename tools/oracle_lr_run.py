#!/usr/bin/env python3
"""The REFERENCE's fit loop (its CPU restatement oracle/dnmf_oracle.py: torch-CPU grid_sample, autograd, torch.optim.Adam,
float64 numpy multiplicative updates -- demo.py:41-46) at a given volume size with the demo's Adam step `lr = 1e-5`
(demo.py:42) and with the step bench.py uses, `1e-5 (50 / size)^2`.  Prints one line per outer iteration: how far the
warps have moved the volume's far corner (voxels), how many frames still have finite coefficients, the largest trace and
the reconstruction loss.  This is test / evidence infrastructure (CPU only, nothing here is shipped):

    python tools/oracle_lr_run.py [size=160] [K=8] [T=16] [outer=3] [epochs=10] [depth=2]

The demo's step is a step in the NORMALISED coefficients of its 50 x 50 x 2 volume: Adam moves every coefficient of every
frame by about `lr` per optimiser step whatever the gradient's size, and a quadratic coefficient multiplies coordinates
up to size^2 -- 0.025 voxels per step at size 50, 0.26 at 160, 2.6 at 512.  profiles/r03_oracle_lr_*.txt hold runs of
this script; tests/test_oracle_lr.py pins the per-step displacement on the oracle.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dnmf_oracle as O  # noqa: E402


def corner_displacement(beta, sz):
    """Largest distance (voxels) over the frames between the warped and the original position of the volume's eight
    corners; NaN coefficients count as infinity."""
    corners = np.array([[x, y, z] for x in (0, sz[0] - 1) for y in (0, sz[1] - 1) for z in (0, sz[2] - 1)], dtype=np.float64)
    basis = O.quadratic_basis(corners.astype(np.float32)).astype(np.float64)       # (8, 10)
    q = np.einsum("ca,adt->cdt", basis, beta.astype(np.float64))                   # (8, 3, T)
    d = np.sqrt(((q - corners[:, :, None]) ** 2).sum(1))
    return float(np.nan_to_num(d, nan=np.inf).max())


def run(size, K, T, outer, epochs, depth, lr, log=print):
    torch.manual_seed(0)
    np.random.seed(0)
    sz = [size, size, depth]
    video, positions, _ = O.generate_video(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    video = np.maximum(video, 0)
    m = O.OracleModel(sz, K, T, positions[:, :, 0], C0=torch.rand(K, T).numpy())
    opt = torch.optim.Adam([m.beta_param], lr=lr)
    batches = [list(range(s, min(T, s + 4))) for s in range(0, T, 4)]
    rows = []
    for it in range(outer):
        t0 = time.time()
        losses = m.update_motion(video, batches, opt, gamma=1, epochs=epochs)
        m.update_footprints(video, 4, gamma_c=0, iter_c=50)
        beta = m.beta
        row = {"outer": it + 1, "steps": (it + 1) * epochs * len(batches), "corner_px": corner_displacement(beta, sz),
               "finite_frames": int(np.isfinite(beta).all(0).all(0).sum()), "frames": T,
               "max_trace": float(np.nan_to_num(m.C, nan=np.inf).max()), "loss_first": losses[0], "loss_last": losses[-1]}
        rows.append(row)
        log(f"lr={lr:.3g} size={size} outer {row['outer']} ({row['steps']} optimiser steps, {time.time() - t0:.0f} s): far corner "
            f"moved {row['corner_px']:.3g} px, finite beta {row['finite_frames']}/{T} frames, max trace {row['max_trace']:.3g}, "
            f"recon loss {row['loss_first']:.3e} -> {row['loss_last']:.3e}")
    return rows


def main():
    a = [int(v) for v in sys.argv[1:]]
    size, K, T, outer, epochs, depth = (a + [160, 8, 16, 3, 10, 2][len(a):])[:6]
    torch.set_num_threads(min(8, torch.get_num_threads()))
    print(f"# oracle fit loop, Simulator video {size}x{size}x{depth}, K={K}, T={T}, batch 4, {epochs} epochs + 50 temporal "
          f"updates per outer iteration", flush=True)
    for lr in (1e-5, 1e-5 * (50.0 / size) ** 2):
        run(size, K, T, outer, epochs, depth, lr, log=lambda s: print(s, flush=True))


if __name__ == "__main__":
    main()
