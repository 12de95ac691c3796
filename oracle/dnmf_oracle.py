"""CPU oracle for the deformable-NMF hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (numpy + torch-CPU) of the algorithm the
reference implements in ``Demix/dNMF.py`` and ``WUtils/Simulator.py``.  It exists so that
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg have something
to compare the HIP path with and to time beside it.  Nothing under ``dnmf_amd/`` imports it:
the product path never routes through this file.

Parity pin: the reference ships no tests or golden data.  The oracle is pinned
by running the reference itself in the build container (``tests/golden/make_golden.py``
imports ``/root/reference`` on CPU) and comparing every function below with the captured
fixtures ``tests/golden/*.npz`` (``tests/test_oracle_golden.py``).

Third-party arithmetic on the path (not in the reference tree; restated here from its
published behaviour and cross-checked against the installed library in the tests):
  * ``torch.nn.functional.grid_sample`` 5-D, mode='bilinear' (trilinear), padding 'zeros',
    align_corners=True (torch 2.10; ATen/native/GridSampler.{h,cpp}) -> ``trilinear_sample``
    and ``trilinear_sample_backward_grid``.
  * ``scipy.interpolate.NearestNDInterpolator`` (cKDTree nearest neighbour) -> used as is.
  * ``sklearn`` GaussianProcessRegressor.sample_y on an un-fitted regressor -> restated in
    ``gp_prior_samples`` (prior mean 0, kernel sigma*RBF(ls), RandomState(0)).
  * ``scipy.sparse.rand`` (draw order of the global numpy RandomState) -> used as is.

Z == 1 ("2-D") volumes are undefined in the reference (0/0 in the grid normalisation,
``Demix/dNMF.py:55``).  Here, and in the HIP path, Z == 1 means "z coordinate pinned to 0":
exactly what a Z == 2 reference run with two identical slices and an identity z-row of beta
computes per slice (fixture G10).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

F32 = np.float32

# --------------------------------------------------------------------------------------
# spatial model (Demix/dNMF.py:18-122)
# --------------------------------------------------------------------------------------


def voxel_lattice(sz):
    """Integer voxel coordinates, shape (X,Y,Z,3) fp32.  Demix/dNMF.py:22."""
    X, Y, Z = (int(s) for s in sz)
    g = np.stack(np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij"), axis=-1)
    return g.astype(F32)


def quadratic_basis(P):
    """[1, x, y, z, x^2, y^2, z^2, xy, xz, yz] per voxel.  Demix/dNMF.py:46-51."""
    P = np.asarray(P, dtype=F32)
    x, y, z = P[..., 0], P[..., 1], P[..., 2]
    one = x * 0 + 1
    return np.stack([one, x, y, z, x * x, y * y, z * z, x * y, x * z, y * z], axis=-1).astype(F32)


def identity_beta(T):
    """(10,3,T) fp32 with rows 1..3 = I.  Demix/dNMF.py:24-26."""
    b = np.zeros((10, 3, int(T)), dtype=F32)
    b[1, 0], b[2, 1], b[3, 2] = 1, 1, 1
    return b


def gaussian_footprints(sz, pos, sigma):
    """A[x,y,z,k] = exp(-sum_d (coord_d - pos[k,d])^2 / sigma_k^2), fp32.  Demix/dNMF.py:39-40.

    Evaluated with torch-CPU fp32 ops in the reference's order (subtract, square, divide by
    sigma^2, sum over d, negate inside exp) so that ``A`` is bit-identical to the reference's.
    """
    lat = torch.from_numpy(voxel_lattice(sz))
    pos = torch.as_tensor(np.asarray(pos, dtype=F32))
    sigma = torch.as_tensor(np.asarray(sigma, dtype=F32))
    d2 = (-(lat[:, :, :, :, None] - pos.T[None, None, None, :, :]) ** 2 / sigma[None, None, None, None, :] ** 2).sum(3)
    return torch.exp(d2).numpy()


def poly_grid(basis, beta_t, sz):
    """Warped coordinates q = basis . beta_t (voxel units) and the normalised grid.

    basis (X,Y,Z,10), beta_t (10,3,B) -> q (X,Y,Z,3,B), n (X,Y,Z,3,B), fp32.
    Demix/dNMF.py:54-55.  The contraction goes through torch.einsum like the reference so the
    fp32 summation order is the library's.  For Z == 1 the z component of ``n`` is set to -1
    (pinned to slice 0) instead of the reference's 0/0.
    """
    q = torch.einsum("mnza,abt->mnzbt", torch.from_numpy(np.ascontiguousarray(basis)),
                     torch.from_numpy(np.ascontiguousarray(beta_t)))
    szm1 = torch.tensor([float(int(s) - 1) for s in sz], dtype=torch.float32)
    if int(sz[2]) == 1:
        den = szm1.clone()
        den[2] = 1.0
        n = 2 * q / den[None, None, None, :, None] - 1
        n[:, :, :, 2, :] = -1.0
    else:
        n = 2 * q / szm1[None, None, None, :, None] - 1
    return q.numpy(), n.numpy()


def _unnormalize(n, size):
    """ATen grid_sampler_unnormalize, align_corners=True: ((n + 1) / 2) * (size - 1), fp32."""
    return ((n.astype(F32) + F32(1)) / F32(2)) * F32(size - 1)


def trilinear_sample(A, n):
    """Zero-padded trilinear gather of all K channels (ATen grid_sampler_3d_cpu semantics).

    A (X,Y,Z,K) fp32, n (X,Y,Z,3,B) normalised coordinates whose component 0 indexes the X
    axis (that is what the permutes at Demix/dNMF.py:56-57 amount to).
    Returns A_t (B,K,X,Y,Z) fp32.
    """
    A = np.asarray(A, dtype=F32)
    X, Y, Z, K = A.shape
    B = n.shape[-1]
    out = np.zeros((B, K) + n.shape[:3], dtype=F32)
    for b in range(B):
        ix = _unnormalize(n[..., 0, b], X)
        iy = _unnormalize(n[..., 1, b], Y)
        iz = _unnormalize(n[..., 2, b], Z)
        x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
        acc = np.zeros(n.shape[:3] + (K,), dtype=F32)
        # corner order of the ATen kernel: z outer (top/bottom), then y (north/south), then x
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    cx, cy, cz = x0 + dx, y0 + dy, z0 + dz
                    wx = (ix - x0) if dx else (x0 + F32(1) - ix)
                    wy = (iy - y0) if dy else (y0 + F32(1) - iy)
                    wz = (iz - z0) if dz else (z0 + F32(1) - iz)
                    w = (wx * wy * wz).astype(F32)
                    ok = (cx >= 0) & (cx <= X - 1) & (cy >= 0) & (cy <= Y - 1) & (cz >= 0) & (cz <= Z - 1)
                    cxi = np.clip(cx, 0, X - 1).astype(np.int64)
                    cyi = np.clip(cy, 0, Y - 1).astype(np.int64)
                    czi = np.clip(cz, 0, Z - 1).astype(np.int64)
                    vals = A[cxi, cyi, czi, :]
                    acc += np.where(ok[..., None], vals * w[..., None], F32(0)).astype(F32)
        out[b] = np.moveaxis(acc, -1, 0)
    return out


def trilinear_sample_torch(A, n):
    """Same gather through the library call the reference makes (Demix/dNMF.py:56-57)."""
    A = torch.from_numpy(np.ascontiguousarray(A, dtype=F32))
    n = torch.from_numpy(np.ascontiguousarray(n, dtype=F32))
    B = n.shape[-1]
    inp = A.permute(3, 2, 1, 0)[None].expand(B, -1, -1, -1, -1)
    out = F.grid_sample(inp, n.permute(4, 2, 1, 0, 3), mode="bilinear", padding_mode="zeros", align_corners=True)
    return out.permute(0, 1, 4, 3, 2).contiguous().numpy()


def log_det_jac(Bm, P):
    """log|det J| of the quadratic map at point P.  Demix/dNMF.py:107-122, kept literally:
    rows 8/9 are used as (yz, xz) here although ``quadratic_basis`` orders them (xz, yz)."""
    Bm = np.asarray(Bm, dtype=F32)
    x, y, z = (F32(P[0]), F32(P[1]), F32(P[2]))
    two = F32(2)
    J = np.empty((3, 3), dtype=F32)
    for c in range(3):
        J[0, c] = Bm[1, c] + two * Bm[4, c] * x + Bm[7, c] * y + Bm[9, c] * z
        J[1, c] = Bm[2, c] + two * Bm[5, c] * y + Bm[7, c] * x + Bm[8, c] * z
        J[2, c] = Bm[3, c] + two * Bm[6, c] * z + Bm[8, c] * y + Bm[9, c] * x
    a, b, c_ = J[0, 0], J[1, 0], J[2, 0]
    d, e, f = J[0, 1], J[1, 1], J[2, 1]
    g, h, i = J[0, 2], J[1, 2], J[2, 2]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c_ * (d * h - e * g)
    return F32(np.log(np.abs(F32(det))))


def corner_reg(beta_t, sz):
    """reg[b] = log_det_jac(beta_b, sz-1)^2 + log_det_jac(beta_b, 0)^2.  Demix/dNMF.py:60-61."""
    sz = np.asarray([int(s) for s in sz], dtype=F32)
    out = np.empty(beta_t.shape[2], dtype=F32)
    for b in range(beta_t.shape[2]):
        out[b] = log_det_jac(beta_t[:, :, b], sz - 1) ** 2 + log_det_jac(beta_t[:, :, b], sz * 0) ** 2
    return out


def forward(A, basis, beta, sz, times, C, sampler=trilinear_sample):
    """ExponentialFP.forward.  Demix/dNMF.py:53-62.

    Returns (A_tC (B,X,Y,Z), A_t (B,K,X,Y,Z), grid (X,Y,Z,3,B) normalised, reg (B,)).
    """
    times = list(times)
    beta_t = np.ascontiguousarray(beta[:, :, times], dtype=F32)
    _, n = poly_grid(basis, beta_t, sz)
    A_t = sampler(A, n)
    Ct = torch.from_numpy(np.ascontiguousarray(np.asarray(C, dtype=F32)[:, times]))
    A_tC = torch.einsum("tkmnz,kt->tmnz", torch.from_numpy(A_t), Ct).numpy()
    return A_tC, A_t, n, corner_reg(beta_t, sz)


def mse_beta_grad_autograd(A, basis, beta, sz, times, C, frames):
    """d mse(A_tC, frames) / d beta through torch autograd on the reference's op sequence
    (einsum -> normalise -> grid_sample -> einsum -> mse_loss).  Demix/dNMF.py:54-58,188-190.
    Returns (loss, grad (10,3,T) fp32) - grad is zero outside ``times``."""
    times = list(times)
    beta_t = torch.tensor(np.asarray(beta, dtype=F32), requires_grad=True)
    A_th = torch.from_numpy(np.ascontiguousarray(A, dtype=F32))
    tr = torch.from_numpy(np.ascontiguousarray(basis, dtype=F32))
    q = torch.einsum("mnza,abt->mnzbt", tr, beta_t[:, :, times])
    if int(sz[2]) == 1:
        den = torch.tensor([float(int(sz[0]) - 1), float(int(sz[1]) - 1), 1.0])
        n = 2 * q / den[None, None, None, :, None] - 1
        n = torch.cat((n[:, :, :, :2, :], torch.full_like(n[:, :, :, 2:, :], -1.0)), 3)
    else:
        den = torch.tensor([float(int(s) - 1) for s in sz])
        n = 2 * q / den[None, None, None, :, None] - 1
    B = len(times)
    A_t = F.grid_sample(A_th.permute(3, 2, 1, 0)[None].expand(B, -1, -1, -1, -1), n.permute(4, 2, 1, 0, 3),
                        align_corners=True).permute(0, 1, 4, 3, 2)
    Ct = torch.from_numpy(np.ascontiguousarray(np.asarray(C, dtype=F32)[:, times]))
    A_tC = torch.einsum("tkmnz,kt->tmnz", A_t, Ct)
    loss = F.mse_loss(A_tC, torch.from_numpy(np.ascontiguousarray(frames, dtype=F32)))
    loss.backward()
    return float(loss.detach()), beta_t.grad.numpy()


def mse_beta_grad_analytic(A, basis, beta, sz, times, C, frames):
    """The same gradient written out by hand (what the fused HIP kernel computes).

    With s(u) = sum_k C[k,t] A[u,k] the reconstruction is the trilinear sample of the single
    image s, so d recon / d q_d = sum_corners (d w_c / d q_d) s(corner) (in-bounds corners only,
    ATen grid_sampler_3d_backward), chained through n = 2 q/(S-1) - 1 (factor 2/(S-1)), the
    un-normalisation (factor (S-1)/2) and q = basis . beta (factor basis[v,a]).
    """
    A = np.asarray(A, dtype=F32)
    X, Y, Z, K = A.shape
    times = list(times)
    B = len(times)
    P = X * Y * Z
    beta_t = np.ascontiguousarray(np.asarray(beta, dtype=F32)[:, :, times])
    _, n = poly_grid(basis, beta_t, sz)
    frames = np.asarray(frames, dtype=F32)
    grad = np.zeros(np.asarray(beta).shape, dtype=F32)
    loss = 0.0
    dims = (X, Y, Z)
    for b, t in enumerate(times):
        s = (A.reshape(P, K).astype(np.float64) @ np.asarray(C, dtype=np.float64)[:, t]).reshape(X, Y, Z)
        u = [_unnormalize(n[..., d, b], dims[d]).astype(np.float64) for d in range(3)]
        f = [np.floor(_unnormalize(n[..., d, b], dims[d])).astype(np.float64) for d in range(3)]
        rec = np.zeros((X, Y, Z))
        dq = [np.zeros((X, Y, Z)) for _ in range(3)]
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    off = (dx, dy, dz)
                    c = [f[d] + off[d] for d in range(3)]
                    ok = np.ones((X, Y, Z), bool)
                    for d in range(3):
                        ok &= (c[d] >= 0) & (c[d] <= dims[d] - 1)
                    ci = [np.clip(c[d], 0, dims[d] - 1).astype(np.int64) for d in range(3)]
                    val = np.where(ok, s[ci[0], ci[1], ci[2]], 0.0)
                    w = [(u[d] - f[d]) if off[d] else (f[d] + 1 - u[d]) for d in range(3)]
                    sg = [1.0 if off[d] else -1.0 for d in range(3)]
                    rec += val * w[0] * w[1] * w[2]
                    dq[0] += val * sg[0] * w[1] * w[2]
                    dq[1] += val * w[0] * sg[1] * w[2]
                    dq[2] += val * w[0] * w[1] * sg[2]
        resid = rec - frames[b]
        loss += float((resid ** 2).sum())
        g = 2.0 * resid / (B * P)
        bas = np.asarray(basis, dtype=np.float64).reshape(P, 10)
        for d in range(3):
            if d == 2 and Z == 1:
                continue
            chain = (dims[d] - 1) / 2.0 * (2.0 / (dims[d] - 1))
            grad[:, d, t] = (bas * (g * dq[d] * chain).reshape(P, 1)).sum(0)
    return loss / (B * P), grad


# --------------------------------------------------------------------------------------
# NMF multiplicative updates (Demix/dNMF.py:139-160)
# --------------------------------------------------------------------------------------


def update_temporal(A_t, C, Y, gamma=None):
    """One multiplicative update of the traces.  Demix/dNMF.py:139-149.

    A_t (X,Y,Z,K,T), C (K,T), Y (X,Y,Z,T); float64 in the reference (numpy einsum)."""
    G = np.einsum("mnzkt,mnzlt->klt", A_t, A_t)
    C1 = np.einsum("mnzkt,mnzt->kt", A_t, Y)
    C2 = np.einsum("klt,lt->kt", G, C)
    if gamma is not None:
        nbr = np.hstack((C[:, :1], C[:, :-1])) + np.hstack((C[:, 1:], C[:, -1:]))
        C1 = C1 + gamma * nbr
        C2 = C2 + 2 * gamma * C
    return C * C1 / (C2 + 1e-32)


def gram_rhs(A_t, Y):
    """The two frame-wise contractions of update_temporal on their own (Demix/dNMF.py:141-142):
    G (K,K,T), r (K,T)."""
    return np.einsum("mnzkt,mnzlt->klt", A_t, A_t), np.einsum("mnzkt,mnzt->kt", A_t, Y)


def mu_temporal_from_gram(G, r, C, gamma=None, iters=1):
    """``iters`` multiplicative updates given the (constant) Gram and rhs: the hoisted form of the
    loop at Demix/dNMF.py:172-173, identical in exact arithmetic."""
    C = np.asarray(C, dtype=np.float64).copy()
    for _ in range(iters):
        C1 = r.copy()
        C2 = np.einsum("klt,lt->kt", G, C)
        if gamma is not None:
            nbr = np.hstack((C[:, :1], C[:, :-1])) + np.hstack((C[:, 1:], C[:, -1:]))
            C1 = C1 + gamma * nbr
            C2 = C2 + 2 * gamma * C
        C = C * C1 / (C2 + 1e-32)
    return C


def update_spatial(A, C, Y_i, D=None, gamma=None):
    """One multiplicative update of (un-warped) footprints.  Demix/dNMF.py:151-160.
    A (m,n,K), C (K,T), Y_i (m,n,T), D None or (m,n,K)."""
    C_s = np.einsum("kt,pt->kp", C, C)
    A1 = np.einsum("mnt,kt->mnk", Y_i, C)
    A2 = np.einsum("mnk,kp->mnp", A, C_s)
    if D is not None:
        A2 = A2 + gamma * D
    return A * A1 / (A2 + 1e-32)


def distance_penalty(sz, positions):
    """D = 1 - exp(-0.01 * ||voxel - pos_k||), (X,Y,Z,K) float64.  Demix/dNMF.py:133-137."""
    lat = voxel_lattice(sz).reshape(-1, 3).astype(np.float64)
    pos = np.asarray(positions, dtype=np.float64)
    d = np.sqrt(((lat[:, None, :] - pos[None, :, :]) ** 2).sum(-1))
    X, Y, Z = (int(s) for s in sz)
    return (1 - np.exp(-0.01 * d)).reshape(X, Y, Z, pos.shape[0])


# --------------------------------------------------------------------------------------
# registration by nearest scatter (Demix/dNMF.py:69-103)
# --------------------------------------------------------------------------------------


def image_iwarp(im, flow, lattice):
    """Nearest-neighbour inverse warp of one frame.  Demix/dNMF.py:95-103."""
    from scipy.interpolate import NearestNDInterpolator
    pts = np.stack([flow[..., 0].reshape(-1), flow[..., 1].reshape(-1), flow[..., 2].reshape(-1)], 1)
    return NearestNDInterpolator(pts, np.asarray(im).reshape(-1))(lattice).reshape(np.asarray(im).shape)


def pushforward_flow(n, sz):
    """Un-normalisation used for the registration only: ((n+1)/2)*sz[d] (not sz[d]-1).
    Demix/dNMF.py:81-83.  n (X,Y,Z,3,B) fp32 -> (X,Y,Z,3,B) fp32."""
    out = np.array(n, dtype=F32, copy=True)
    for d in range(3):
        out[:, :, :, d] = ((out[:, :, :, d] + F32(1)) / F32(2)) * F32(int(sz[d]))
    return out


# --------------------------------------------------------------------------------------
# orchestration (Demix/dNMF.py:124-194)
# --------------------------------------------------------------------------------------


class OracleModel:
    """State of ExponentialFP + DeformableNMF on the CPU (Demix/dNMF.py:19-43, 126-137)."""

    def __init__(self, sz, K, T, positions, shape_std=3.0, C0=None):
        self.sz = [int(s) for s in sz]
        self.K, self.T = int(K), int(T)
        self.lattice = voxel_lattice(self.sz)
        self.basis = quadratic_basis(self.lattice)
        self.beta_param = torch.tensor(identity_beta(T), requires_grad=True)  # leaf, like fp.beta
        self.sigma = np.full(self.K, shape_std, dtype=F32)
        self.pos = np.asarray(positions, dtype=F32)
        self.A = gaussian_footprints(self.sz, self.pos, self.sigma)
        self.C = np.asarray(C0, dtype=F32) if C0 is not None else torch.rand(K, T).numpy()

    @property
    def beta(self):
        """(10,3,T) fp32 numpy view of the leaf tensor the caller's optimiser steps."""
        return self.beta_param.detach().numpy()

    def forward(self, times, C=None, sampler=trilinear_sample):
        return forward(self.A, self.basis, self.beta, self.sz, times, self.C if C is None else C, sampler)

    def pushforward(self, video, batch_size, with_registration=False):
        """spatial_pushforward.  Demix/dNMF.py:69-93.  video (X,Y,Z,T) fp32, frames >= 0.
        Returns A_t (X,Y,Z,K,T) f64, Y_i (X,Y,Z,T) f64 (zeros unless with_registration), Y."""
        X, Y_, Z = self.sz
        T = video.shape[3]
        A_t = np.zeros((X, Y_, Z, self.K, T))
        Yi = np.zeros((X, Y_, Z, T))
        Yv = np.zeros((X, Y_, Z, T))
        lat = self.lattice.astype(np.int64)
        for s in range(0, T, batch_size):
            times = list(range(s, min(T, s + batch_size)))
            _, a_t, n, _ = self.forward(times, sampler=trilinear_sample_torch)
            A_t[..., times] = np.transpose(a_t, [2, 3, 4, 1, 0])
            Yv[..., times] = video[..., times]
            if with_registration:
                flow = pushforward_flow(n, self.sz)
                for b, t in enumerate(times):
                    Yi[..., t] = image_iwarp(video[..., t], flow[..., b], lat)
        return A_t, Yi, Yv

    def update_footprints(self, video, batch_size, gamma_c=1e-2, iter_c=10, with_registration=False):
        """Demix/dNMF.py:163-179 (Gram/rhs recomputed every iteration, like the reference)."""
        A_t, Yi, Yv = self.pushforward(video, batch_size, with_registration)
        C = self.C.astype(np.float64)
        for _ in range(iter_c):
            C = update_temporal(A_t, C, Yv, gamma=gamma_c)
        self.C = C.astype(F32)
        return A_t, Yi, Yv

    def update_motion(self, video, batches, optimizer, gamma=0, epochs=1):
        """Demix/dNMF.py:181-194 with an explicit batch list (list of lists of frame indices, the
        same list every epoch); ``optimizer`` is the caller's, built on ``self.beta_param``
        (demo.py:42).  Returns the per-batch reconstruction losses."""
        beta = self.beta_param
        A_th = torch.from_numpy(self.A)
        tr = torch.from_numpy(self.basis)
        Z1 = self.sz[2] == 1
        den = torch.tensor([float(self.sz[0] - 1), float(self.sz[1] - 1), 1.0 if Z1 else float(self.sz[2] - 1)])
        losses = []
        for _ in range(epochs):
            for times in batches:
                times = list(times)
                optimizer.zero_grad()
                q = torch.einsum("mnza,abt->mnzbt", tr, beta[:, :, times])
                n = 2 * q / den[None, None, None, :, None] - 1
                if Z1:
                    n = torch.cat((n[:, :, :, :2, :], torch.full_like(n[:, :, :, 2:, :], -1.0)), 3)
                B = len(times)
                A_t = F.grid_sample(A_th.permute(3, 2, 1, 0)[None].expand(B, -1, -1, -1, -1),
                                    n.permute(4, 2, 1, 0, 3), align_corners=True).permute(0, 1, 4, 3, 2)
                A_tC = torch.einsum("tkmnz,kt->tmnz", A_t, torch.from_numpy(np.ascontiguousarray(self.C[:, times])))
                frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(np.asarray(video, dtype=F32)[..., times], -1, 0)))
                recon = F.mse_loss(A_tC, frames)
                reg = torch.from_numpy(corner_reg(beta.detach().numpy()[:, :, times], self.sz))
                loss = recon + gamma * reg.mean()
                loss.backward()
                optimizer.step()
                losses.append(float(recon.detach()))
        return losses


# --------------------------------------------------------------------------------------
# synthetic video (WUtils/Simulator.py:20-77, 174-212, 362-391)
# --------------------------------------------------------------------------------------


def gp_prior_samples(x, sigma, ls, T):
    """sample_y of an un-fitted GaussianProcessRegressor(kernel=sigma*RBF(ls)) with its default
    random_state=0: T draws from N(0, sigma*exp(-d^2/(2 ls^2))).  WUtils/Simulator.py:380-390.
    Returns (len(x), T)."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    d2 = (x - x.T) ** 2
    cov = sigma * np.exp(-0.5 * d2 / (ls * ls))
    rng = np.random.RandomState(0)
    return rng.multivariate_normal(np.zeros(len(x)), cov, int(T)).T


def generate_gp_motion(K, T, sigma, ls, sz):
    """WUtils/Simulator.py:362-391.  Consumes K*3 draws of the global numpy RandomState."""
    A0 = np.random.rand(K, 3) * np.array([int(s) for s in sz])
    S = np.array([A0[:, d][:, None] + gp_prior_samples(A0[:, d], sigma[d], ls[d], T) for d in range(3)]).T
    return S.transpose(1, 2, 0).astype(F32)  # (K,3,T)


def simulate_exponential_traces(K, T, density=0.1, b=1):
    """WUtils/Simulator.py:174-195."""
    from scipy.sparse import rand as sprand
    traces = b + 0 * np.random.rand(K, T)
    kernel = np.exp(np.arange(0, -3, -0.3))
    for k in range(K):
        a = sprand(1, T + len(kernel) - 1, density=density, format="csr")
        a.data[:] = 1
        traces[k, :] += np.convolve(np.array(a.todense()).flatten(), kernel, "valid")
    return traces


def render_cell(sz, mean, shape_std, amp):
    """One neuron's contribution to one frame: amp * exp(-r^2 / (2 shape_std)) over the WHOLE
    volume, float64 (multivariate_normal(mean, shape_std*I).pdf times its normaliser).
    WUtils/Simulator.py:70-73, 197-212."""
    lat = voxel_lattice(sz).astype(np.float64)
    r2 = ((lat - np.asarray(mean, dtype=np.float64)) ** 2).sum(-1)
    return amp * np.exp(-0.5 * r2 / shape_std)


def render_video(positions, traces, sz, shape_std, noise):
    """The render loop + normalisation of generate_video given its random inputs.
    WUtils/Simulator.py:66-77.  positions (K,3,T) fp32, traces (K,T) f64, noise (X,Y,Z,T) fp32
    (already multiplied by bg_std).  Per (t,k) the f64 patch is cast to fp32 and added in fp32."""
    K, _, T = positions.shape
    X, Y, Z = (int(s) for s in sz)
    video = torch.zeros(X, Y, Z, T)
    for t in range(T):
        for k in range(K):
            patch = render_cell(sz, positions[k, :, t], shape_std, traces[k, t])
            video[:, :, :, t] = video[:, :, :, t] + torch.tensor(patch).float()
    video /= (video ** 2).sum()
    video += torch.as_tensor(noise)
    return (video / video.max()).numpy()


def generate_video(K, T, sz, shape_std, density, bg_snr, motion_par):
    """generate_video(traces='exp', motion='gp').  WUtils/Simulator.py:20-77.  Random draw order:
    numpy global (positions, traces) then torch global (noise)."""
    positions = generate_gp_motion(K, T, motion_par["sigma"], motion_par["ls"], sz)
    traces = simulate_exponential_traces(K, T, density)
    bg_std = np.sqrt(10 ** (bg_snr / 10))
    X, Y, Z = (int(s) for s in sz)
    noise = bg_std * torch.distributions.normal.Normal(0, 1).sample(np.array([X, Y, Z, T]))
    return render_video(positions, traces, sz, shape_std, noise), positions, traces
