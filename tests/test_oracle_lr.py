"""Why bench.py does not use the demo's Adam step on a 512 x 512 volume, pinned on the REFERENCE side (CPU, the oracle's
restatement of demo.py:41-46 with torch.optim.Adam): the demo's `lr = 1e-5` (demo.py:42) is a step in the normalised
coefficients of a 50 x 50 volume.  Adam's first step moves every coefficient of every frame of the mini-batch by exactly
`lr` (m / sqrt(v) = sign(g)), the quadratic coefficients multiply coordinates up to size^2, so one optimiser step displaces
the far corner of the volume by about lr * size^2 voxels per quadratic term: 0.025 px at size 50, 0.26 px at 160, 2.6 px
-- a footprint width -- at 512.  tools/oracle_lr_run.py runs the whole loop (profiles/r03_oracle_lr_*.txt)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_first_adam_step_moves_every_coefficient_by_lr_and_the_corner_by_lr_size_squared():
    from oracle import dnmf_oracle as O
    from oracle_lr_run import corner_displacement
    disp = {}
    for size in (50, 160):
        torch.manual_seed(0)
        np.random.seed(0)
        sz, K, T = [size, size, 2], 4, 4
        video, positions, _ = O.generate_video(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
        video = np.maximum(video, 0)
        m = O.OracleModel(sz, K, T, positions[:, :, 0], C0=torch.rand(K, T).numpy())
        beta0 = m.beta.copy()
        opt = torch.optim.Adam([m.beta_param], lr=1e-5)
        m.update_motion(video, [list(range(T))], opt, gamma=1, epochs=1)     # ONE optimiser step
        step = np.abs(m.beta - beta0)
        # Adam's first step: lr * g / (|g| + eps) -- exactly lr wherever the gradient is not tiny, never more
        assert step.max() <= 1.01e-5      # (an increment of a coefficient near 1 is rounded to its fp32 spacing)
        assert np.median(step[[4, 5, 7]][:, :2]) > 0.99e-5          # the quadratic coefficients of x and y moved by lr
        disp[size] = corner_displacement(m.beta, sz)
    # the far corner moves by up to lr * (3 quadratic terms) * size^2 (+ lr * 2 size + lr): ~0.05 px at 50, ~0.5 px at 160
    assert disp[50] < 0.2 and disp[160] > 0.3
    assert 6.0 < disp[160] / disp[50] < 14.0                        # ~ (160 / 50)^2 = 10.2
    # at 512 the same step is lr * 512^2 = 2.6 px per quadratic term: bench.py scales lr by (50 / size)^2
    assert 1e-5 * 512 ** 2 > 2.5
