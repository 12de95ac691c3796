"""Synthetic video generator with the semantics of the reference's ``WUtils/Simulator.py``
(``generate_video`` :20-77, ``generate_gp_motion`` :362-391, ``simulate_exponential_traces`` :174-195,
``simulate_cell`` :197-212) -- the measurement-input contract of the hot path.

Same random-draw order as the reference (numpy global state for positions and traces, torch global state
for the background noise), so a seeded call reproduces the reference video.  Differences, none of which
changes a value:
  * a neuron is rendered on the window where its fp32 value can be non-zero instead of the whole volume
    (the reference evaluates a full-volume float64 pdf, casts it to fp32 and adds it: everything further
    than ~25 px from the centre is exactly 0 after the cast);
  * the video is stored frame-major, ``(T,X,Y,Z)`` contiguous, and exposed as the reference's
    ``(X,Y,Z,T)`` view;
  * only ``motion='gp'`` and ``traces='exp'`` exist: the reference's other motion models raise before
    producing anything.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def simulate_exponential_traces(K, T, density=.1, b=1):
    """Baseline ``b`` plus sparse unit spikes convolved with a 10-tap decaying exponential."""
    from scipy.sparse import rand as sparse_rand
    traces = b + 0 * np.random.rand(K, T)      # the draw is part of the reference's RNG stream
    kernel = np.exp(np.arange(0, -3, -.3))
    for k in range(K):
        spikes = sparse_rand(1, T + len(kernel) - 1, density=density, format='csr')
        spikes.data[:] = 1
        traces[k, :] += np.convolve(np.asarray(spikes.todense()).ravel(), kernel, 'valid')
    return traces


def generate_gp_motion(K, T=100, sigma=(10, 10, 10), ls=(10, 10, 10), sz=(10, 10, 1)):
    """Positions (K,3,T): a random start inside the volume plus, per axis, T draws of a zero-mean Gaussian
    process over the start coordinates with kernel sigma*RBF(ls).  The reference samples an un-fitted
    sklearn GaussianProcessRegressor, whose ``sample_y`` seeds its own RandomState(0)."""
    start = np.random.rand(K, 3) * np.array([int(s) for s in sz])
    out = np.empty((T, K, 3))
    for d in range(3):
        x = start[:, d][:, None]
        cov = sigma[d] * np.exp(-0.5 * (x - x.T) ** 2 / (ls[d] * ls[d]))
        draws = np.random.RandomState(0).multivariate_normal(np.zeros(K), cov, int(T)).T   # (K,T)
        out[:, :, d] = (x + draws).T
    return torch.tensor(out.transpose(1, 2, 0)).float()


def _render_window_radius(shape_std, amp_max):
    # amp*exp(-r^2/(2 s)) rounds to 0 in fp32 below 2^-150
    return int(math.sqrt(2.0 * shape_std * (math.log(max(amp_max, 1.0)) + 150 * math.log(2.0)))) + 2


def render_frames(positions, traces, sz, shape_std, out=None):
    """Sum of isotropic Gaussians exp(-r^2/(2*shape_std)) scaled by the traces, frame-major (T,X,Y,Z) fp32.
    Per neuron the float64 window is cast to fp32 and added in fp32, in neuron order, like the reference."""
    X, Y, Z = (int(s) for s in sz)
    pos = np.asarray(positions, dtype=np.float64)
    K, _, T = pos.shape
    tr = np.asarray(traces, dtype=np.float64)
    video = np.zeros((T, X, Y, Z), dtype=np.float32) if out is None else out
    R = _render_window_radius(float(shape_std), float(tr.max()))
    gx, gy, gz = np.arange(X, dtype=np.float64), np.arange(Y, dtype=np.float64), np.arange(Z, dtype=np.float64)
    inv = 0.5 / float(shape_std)
    for t in range(T):
        frame = video[t]
        for k in range(K):
            cx, cy, cz = pos[k, :, t]
            x0, x1 = max(0, int(math.floor(cx)) - R), min(X, int(math.floor(cx)) + R + 2)
            y0, y1 = max(0, int(math.floor(cy)) - R), min(Y, int(math.floor(cy)) + R + 2)
            z0, z1 = max(0, int(math.floor(cz)) - R), min(Z, int(math.floor(cz)) + R + 2)
            if x0 >= x1 or y0 >= y1 or z0 >= z1:
                continue
            r2 = ((gx[x0:x1] - cx) ** 2)[:, None, None] + ((gy[y0:y1] - cy) ** 2)[None, :, None] \
                + ((gz[z0:z1] - cz) ** 2)[None, None, :]
            frame[x0:x1, y0:y1, z0:z1] += (tr[k, t] * np.exp(-inv * r2)).astype(np.float32)
    return video


def generate_video(K, T, sz=[20, 20, 1], shape_std=3, density=.1, bg_snr=-1, traces='exp', motion='sq',
                   motion_par={'means': [.0, .0, .0], 'snr': [-3, -3, -3]}):
    """Simulate a video of active, moving Gaussian neurons.

    Returns ``video`` torch (X,Y,Z,T) fp32 (a view of frame-major storage), ``positions`` torch (K,3,T) fp32,
    ``traces`` numpy (K,T) float64 -- the reference's return triple."""
    if motion != 'gp':
        raise NotImplementedError("only motion='gp' is available (the reference's 'sq', 'q' and 'qs' models "
                                  "raise before returning)")
    if traces != 'exp':
        raise NotImplementedError("only traces='exp' is available")
    positions = generate_gp_motion(K, T, motion_par['sigma'], motion_par['ls'], sz)
    tr = simulate_exponential_traces(K, T, density)
    X, Y, Z = (int(s) for s in sz)
    bg_std = np.sqrt(10 ** (bg_snr / 10))
    # drawn before the render loop and in the reference's (X,Y,Z,T) element order
    noise = bg_std * torch.distributions.normal.Normal(0, 1).sample(np.array([X, Y, Z, T]))
    frames = torch.from_numpy(render_frames(positions.numpy(), tr, sz, shape_std))
    frames /= (frames ** 2).sum()
    frames += noise.permute(3, 0, 1, 2)
    frames /= frames.max()
    return frames.permute(1, 2, 3, 0), positions, tr


def generate_video_resident(K, T, sz, shape_std=3, density=.1, bg_snr=-1, motion_par=None, device='cuda',
                            t0=0, t1=None, noise=None, group=None):
    """``generate_video(traces='exp', motion='gp')`` rendered on the GPU, frame-major and resident.

    Positions and traces are drawn on the host exactly like ``generate_video`` (same numpy draws for the same
    seed, for all T frames); frames ``t0..t1-1`` are rendered by ``dnmf_render_frames`` into a ``(t1-t0, P)``
    fp32 CUDA tensor.  ``noise`` (X,Y,Z,T) or (T,P): background noise to add (already scaled); None draws
    ``bg_std * N(0,1)`` with the device generator (a different stream from the reference's CPU draw).
    ``group``: torch.distributed group when the T axis is sharded over ranks -- the two global
    normalisers (sum of squares, maximum) are all-reduced so every rank holds its slice of ONE video.
    Returns ``frames (t1-t0, P)``, ``positions (K,3,T)`` torch fp32 (host), ``traces (K,T)`` numpy float64."""
    t1 = T if t1 is None else t1
    positions = generate_gp_motion(K, T, motion_par['sigma'], motion_par['ls'], sz)
    tr = simulate_exponential_traces(K, T, density)
    if torch.device(device).type == 'cpu':   # host renderer: only for the multi-process CPU tests
        frames = torch.from_numpy(render_frames(positions.numpy()[:, :, t0:t1], tr[:, t0:t1], sz, shape_std))
        frames = frames.reshape(t1 - t0, -1)
    else:
        from .. import ops
        frames = ops.render_frames(positions.to(device).contiguous(), torch.from_numpy(tr).to(device).contiguous(),
                                   sz, shape_std, t0, t1 - t0)
    energy = (frames.double() ** 2).sum()
    if group is not None:
        torch.distributed.all_reduce(energy, group=group)
    frames /= energy.float()
    bg_std = float(np.sqrt(10 ** (bg_snr / 10)))
    if noise is None:
        frames += bg_std * torch.randn(frames.shape, device=device)
    else:
        noise = torch.as_tensor(noise)
        if noise.dim() == 4:
            noise = noise.permute(3, 0, 1, 2).reshape(noise.shape[3], -1)
        frames += noise[t0:t1].to(device)
    peak = frames.max()
    if group is not None:
        torch.distributed.all_reduce(peak, op=torch.distributed.ReduceOp.MAX, group=group)
    frames /= peak
    return frames, positions, tr
