"""Thin torch-tensor front end of the C ABI: validates device/dtype/contiguity, passes raw pointers and
the current HIP stream.  torch is plumbing here (device memory, streams); all arithmetic happens in
libdnmf_hip.so."""
from __future__ import annotations

import torch

from . import _lib


# bench.py sets TIMING = {} to collect (start, end) HIP events around the named kernels, on their stream
TIMING = None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class _timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if TIMING is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if TIMING is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            TIMING.setdefault(self.name, []).append((self.a, b))


def _f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ValueError(f"{name}: need a contiguous float32 CUDA tensor, got {t.dtype} {t.device} "
                         f"contiguous={t.is_contiguous()}")
    return t


def _i32(t, device) -> torch.Tensor:
    if isinstance(t, torch.Tensor):
        return t.to(device=device, dtype=torch.int32).contiguous()
    return torch.tensor(list(t), dtype=torch.int32, device=device)


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def build_stamp() -> dict:
    """{source file: hash} the loaded library was compiled from (``dnmf_build_stamp``; dnmf_amd/build.py makes it)."""
    text = _lib.load().dnmf_build_stamp().decode()
    return dict(item.split(":", 1) for item in text.split(";") if ":" in item)


def padded_k(K: int) -> int:
    return _lib.load().dnmf_padded_k(int(K))


HALO = 2   # DNMF_HALO of include/dnmf_hip.h


def halo_voxels(sz) -> int:
    """Floats of one halo-layout image (reconstruction image S, neuron-major footprint At) of the volume ``sz``."""
    X, Y, Z = (int(s) for s in sz)
    return _lib.load().dnmf_halo_voxels(X, Y, Z)


def halo_interior(img: torch.Tensor, sz) -> torch.Tensor:
    """(..., >= halo_voxels) rows in the halo layout -> (..., X, Y, Z) view of the voxels without the border."""
    X, Y, Z = (int(s) for s in sz)
    row = _lib.load().dnmf_halo_row(Y, Z)
    v = img[..., :(X + 2 * HALO) * row].unflatten(-1, (X + 2 * HALO, row))   # views all the way
    return v[..., HALO:HALO + X, HALO * Z:(HALO + Y) * Z].unflatten(-1, (Y, Z))


def pack_footprints(A: torch.Tensor) -> torch.Tensor:
    """A (..., K) -> packed (P, Kp) zero-padded copy the Gram / recon kernels read."""
    K = A.shape[-1]
    A2 = _f32(A.reshape(-1, K), "A")
    Kp = padded_k(K)
    out = torch.empty((A2.shape[0], Kp), dtype=torch.float32, device=A.device)
    _lib.check(_lib.load().dnmf_pack_footprints(A2.data_ptr(), A2.shape[0], K, out.data_ptr(), Kp, _stream()),
               "dnmf_pack_footprints")
    return out


def warp_gather(A: torch.Tensor, beta: torch.Tensor, times, want_A_t=True, want_grid=True):
    """K1.  A (X,Y,Z,K), beta (10,3,T) -> A_t (B,K,X,Y,Z), grid (X,Y,Z,3,B)."""
    X, Y, Z, K = A.shape
    _f32(A, "A"), _f32(beta, "beta")
    tt = _i32(times, A.device)
    B = tt.numel()
    A_t = torch.empty((B, K, X, Y, Z), dtype=torch.float32, device=A.device) if want_A_t else None
    grid = torch.empty((X, Y, Z, 3, B), dtype=torch.float32, device=A.device) if want_grid else None
    _lib.check(_lib.load().dnmf_warp_gather(A.data_ptr(), X, Y, Z, K, beta.data_ptr(), beta.shape[2], tt.data_ptr(), B,
                                            _ptr(A_t), _ptr(grid), _stream()), "dnmf_warp_gather")
    return A_t, grid


def recon_image(Apk: torch.Tensor, K: int, sz, C: torch.Tensor, times, out: torch.Tensor | None = None) -> torch.Tensor:
    """S[b] = sum_k C[k,times[b]] A[:,k] in the halo layout.  Apk (P,Kp), C (K,T) -> (B, halo_voxels(sz))."""
    X, Y, Z = (int(s) for s in sz)
    _f32(Apk, "Apk"), _f32(C, "C")
    P, Kp = Apk.shape
    if P != X * Y * Z:
        raise ValueError(f"recon_image: Apk has {P} rows, the volume {X}x{Y}x{Z} has {X * Y * Z} voxels")
    tt = _i32(times, Apk.device)
    B = tt.numel()
    lds = halo_voxels(sz)
    if out is None:
        out = torch.empty((B, lds), dtype=torch.float32, device=Apk.device)
    if out.shape[0] < B or out.stride(0) < lds or out.stride(1) != 1:
        raise ValueError("recon_image: out must be (>=B, ld) with ld >= halo_voxels(sz)")
    for s in range(0, B, 32768):   # frames ride on gridDim.y
        n = min(32768, B - s)
        _lib.check(_lib.load().dnmf_recon_image(Apk.data_ptr(), X, Y, Z, K, Kp, C.data_ptr(), C.stride(0),
                                                tt[s:].data_ptr(), n, out[s:].data_ptr(), out.stride(0), _stream()),
                   "dnmf_recon_image")
    return out


def recon_image_lists(layout, K: int, sz, C: torch.Tensor, times, out: torch.Tensor | None = None,
                      skip_empty: bool = False) -> torch.Tensor:
    """S as ``recon_image`` from the K3n layout (``pack_footprints_lists``): compact footprints, static tile lists.
    ``skip_empty``: tiles without a neuron are left alone -- only for an ``out`` whose rows an earlier call with the same
    layout and ``skip_empty=False`` has written (they hold zeros there)."""
    X, Y, Z = (int(s) for s in sz)
    _f32(C, "C")
    tt = _i32(times, C.device)
    B = tt.numel()
    lds = halo_voxels(sz)
    if out is None:
        out = torch.empty((B, lds), dtype=torch.float32, device=C.device)
    if out.shape[0] < B or out.stride(0) < lds or out.stride(1) != 1:
        raise ValueError("recon_image_lists: out must be (>=B, ld) with ld >= halo_voxels(sz)")
    with _timed("recon_image_lists"):
        rc = _lib.load().dnmf_recon_image_lists_ex(layout["At"].data_ptr(), layout["bbox"].data_ptr(), K, X, Y, Z,
                                                   C.data_ptr(), C.stride(0), tt.data_ptr(), B, out.data_ptr(),
                                                   out.stride(0), 1 if skip_empty else 0, _stream())
    _lib.check(rc, "dnmf_recon_image_lists_ex")
    return out


def warp_recon_grad(S, s_ids, frames, frame_ids, sz, beta, times, grad=None, gout=None, want_recon=False,
                    want_loss=True, want_reg=True, workspace=None, norm_frames=0):
    """K2.  S (>=B, lds) recon images in the halo layout, frames (>=B, ldf) or None with gout (B,P).
    Returns dict(recon, loss, frame_loss, reg); ``grad`` (10,3,T) is incremented in place."""
    X, Y, Z = (int(s) for s in sz)
    P = X * Y * Z
    dev = beta.device
    _f32(beta, "beta")
    tt = _i32(times, dev)
    B = tt.numel()
    lib = _lib.load()
    need = lib.dnmf_warp_recon_grad_workspace(X, Y, Z, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    recon = torch.empty((B, P), dtype=torch.float32, device=dev) if want_recon else None
    loss = torch.empty((1,), dtype=torch.float32, device=dev) if want_loss else None
    frame_loss = torch.empty((B,), dtype=torch.float32, device=dev) if want_loss else None
    reg = torch.empty((B,), dtype=torch.float32, device=dev) if want_reg else None
    sid = _i32(s_ids, dev) if s_ids is not None else None
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    if frames is not None and (frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda):
        raise ValueError("warp_recon_grad: frames must be float32 CUDA with unit inner stride")
    if gout is not None:
        _f32(gout, "gout")
    with _timed("warp_recon_grad"):
        rc = lib.dnmf_warp_recon_grad(
            S.data_ptr(), S.stride(0), _ptr(sid), _ptr(frames), 0 if frames is None else frames.stride(0), _ptr(fid),
            _ptr(gout), X, Y, Z, beta.data_ptr(), beta.shape[2], tt.data_ptr(), B, int(norm_frames), _ptr(recon),
            _ptr(grad),
            _ptr(loss), _ptr(frame_loss), _ptr(reg), workspace.data_ptr(),
            workspace.numel() * workspace.element_size(), _stream())
    _lib.check(rc, "dnmf_warp_recon_grad")
    return {"recon": recon, "loss": loss, "frame_loss": frame_loss, "reg": reg, "workspace": workspace}


def motion_grad_lists(layout, K, sz, C, frames, frame_ids, beta, times, grad, norm_frames, chunk, want=False, workspace=None):
    """The motion gradient of the frames ``times`` from the K3n layout: list reconstruction and K2 alternate over pieces
    of ``chunk`` frames whose images share one buffer (they stay in the Infinity Cache).  ``grad`` (10,3,T) is
    incremented.  Returns dict(frame_loss, reg, workspace) (the first two None unless ``want``)."""
    X, Y, Z = (int(s) for s in sz)
    dev = beta.device
    _f32(beta, "beta"), _f32(C, "C"), _f32(grad, "grad")
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("motion_grad_lists: frames must be float32 CUDA with unit inner stride")
    tt = _i32(times, dev)
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = tt.numel()
    lib = _lib.load()
    chunk = max(1, min(int(chunk), B))
    need = lib.dnmf_motion_grad_lists_workspace(X, Y, Z, chunk, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    frame_loss = torch.empty((B,), dtype=torch.float32, device=dev) if want else None
    reg = torch.empty((B,), dtype=torch.float32, device=dev) if want else None
    with _timed("motion_grad_lists"):
        rc = lib.dnmf_motion_grad_lists(layout["At"].data_ptr(), layout["bbox"].data_ptr(), K, C.data_ptr(), C.stride(0),
                                        frames.data_ptr(), frames.stride(0), _ptr(fid), X, Y, Z, beta.data_ptr(),
                                        beta.shape[2], tt.data_ptr(), B, int(norm_frames), grad.data_ptr(), _ptr(frame_loss),
                                        _ptr(reg), chunk, workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                        _stream())
    _lib.check(rc, "dnmf_motion_grad_lists")
    return {"frame_loss": frame_loss, "reg": reg, "workspace": workspace}


def warp_gram_rhs(Apk, K, sz, beta, times, frames, frame_ids=None, a_frame_stride=0, workspace=None, bf16=False):
    """K3 (``bf16=True``: K3b, operands rounded to bf16, fp32 accumulate).  Returns G (B,K,K), r (B,K) for the
    frames listed.  ``beta`` None: no warp (each voxel's own footprint row, weight 1)."""
    X, Y, Z = (int(s) for s in sz)
    P = X * Y * Z
    dev = Apk.device
    _f32(Apk, "Apk")
    if beta is not None:
        _f32(beta, "beta")
    lib = _lib.load()
    tt = _i32(times, dev) if times is not None else None
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = tt.numel() if tt is not None else (fid.numel() if fid is not None else frames.shape[0])
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("warp_gram_rhs: frames must be float32 CUDA with unit inner stride")
    need = lib.dnmf_warp_gram_rhs_workspace(P, K, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    G = torch.empty((B, K, K), dtype=torch.float32, device=dev)
    r = torch.empty((B, K), dtype=torch.float32, device=dev)
    with _timed("warp_gram_rhs_bf16" if bf16 else "warp_gram_rhs"):
        rc = (lib.dnmf_warp_gram_rhs_bf16 if bf16 else lib.dnmf_warp_gram_rhs)(
            Apk.data_ptr(), Apk.shape[-1], K, a_frame_stride, X, Y, Z, _ptr(beta), B if beta is None else beta.shape[2],
            _ptr(tt), B, frames.data_ptr(), frames.stride(0), _ptr(fid), G.data_ptr(), r.data_ptr(), workspace.data_ptr(),
            workspace.numel() * workspace.element_size(), _stream())
    _lib.check(rc, "dnmf_warp_gram_rhs")
    return G, r, workspace


def mu_temporal(G, r, C, iters: int, nbr=None):
    """K4 without the neighbour term: C (K,T) fp32 updated in place.  ``nbr`` (K,NN) int32: the columns of G that
    can be non-zero per row (``pack_footprints_lists``) -- same result, NN instead of K terms per row."""
    _f32(G, "G"), _f32(r, "r")
    if not (C.is_cuda and C.dtype == torch.float32 and C.stride(1) == 1):
        raise ValueError("mu_temporal: C must be float32 CUDA with unit inner stride")
    T, K = r.shape
    if nbr is not None:
        _lib.check(_lib.load().dnmf_mu_temporal_nbr(G.data_ptr(), r.data_ptr(), C.data_ptr(), C.stride(0), K, T, int(iters),
                                                    nbr.data_ptr(), nbr.shape[1], _stream()), "dnmf_mu_temporal_nbr")
        return C
    _lib.check(_lib.load().dnmf_mu_temporal(G.data_ptr(), r.data_ptr(), C.data_ptr(), C.stride(0), K, T, int(iters),
                                            _stream()), "dnmf_mu_temporal")
    return C


def mu_temporal_step(G, r, Cin, Cout, gamma: float, c_left=None, c_right=None):
    """K4, one round with the neighbour term; Cin/Cout (K,T) float64."""
    _f32(G, "G"), _f32(r, "r")
    for t, n in ((Cin, "Cin"), (Cout, "Cout")):
        if not (t.is_cuda and t.dtype == torch.float64 and t.stride(1) == 1):
            raise ValueError(f"mu_temporal_step: {n} must be float64 CUDA with unit inner stride")
    T, K = r.shape
    _lib.check(_lib.load().dnmf_mu_temporal_step(G.data_ptr(), r.data_ptr(), Cin.data_ptr(), Cout.data_ptr(),
                                                 Cin.stride(0), K, T, float(gamma), _ptr(c_left), _ptr(c_right),
                                                 _stream()), "dnmf_mu_temporal_step")
    return Cout


def render_frames(positions, traces, sz, shape_std, t0=0, T=None, out=None):
    """Simulator render loop on the GPU: frames t0..t0+T-1 of the video as (T,P) fp32."""
    X, Y, Z = (int(s) for s in sz)
    P = X * Y * Z
    pos = _f32(positions, "positions")
    if not (traces.is_cuda and traces.dtype == torch.float64 and traces.is_contiguous()):
        raise ValueError("render_frames: traces must be a contiguous float64 CUDA tensor")
    K, _, T_total = pos.shape
    T = T_total - t0 if T is None else int(T)
    if out is None:
        out = torch.empty((T, P), dtype=torch.float32, device=pos.device)
    lib = _lib.load()
    amp = float(traces.max())
    for s in range(0, T, 32768):
        n = min(32768, T - s)
        _lib.check(lib.dnmf_render_frames(pos.data_ptr(), traces.data_ptr(), K, T_total, t0 + s, n, X, Y, Z,
                                          float(shape_std), amp, out[s:].data_ptr(), out.stride(0), _stream()),
                   "dnmf_render_frames")
    return out


def adam_epoch(beta, grad, exp_avg, exp_avg_sq, step0, frame_step, nsteps, lr, betas, eps, phase, order=None):
    """One phase of the per-column Adam epoch (see include/dnmf_hip.h); tensors updated in place.  ``order`` (T)
    int32: the frames sorted by ``frame_step`` (speed only)."""
    for t, n in ((beta, "beta"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _f32(t, n)
    if grad is not None:
        _f32(grad, "grad")
    fs = _i32(frame_step, beta.device)
    od = _i32(order, beta.device) if order is not None else None
    ws = torch.empty((4 * int(nsteps),), dtype=torch.float32, device=beta.device)
    _lib.check(_lib.load().dnmf_adam_epoch(beta.data_ptr(), _ptr(grad), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                           beta.shape[2], int(step0), fs.data_ptr(), _ptr(od), int(nsteps), float(lr),
                                           float(betas[0]), float(betas[1]), float(eps), int(phase), ws.data_ptr(),
                                           ws.numel() * 4, _stream()),
               "dnmf_adam_epoch")


def spatial_accum(Y, C, frame_ids=None, times=None, A1=None, Cs=None, accumulate=False):
    """K5.  Y (>=T, ldy) frames, C (K, ldc) traces -> A1 (P,K) = Y^T C^T over the frames, Cs (K,K) = C C^T."""
    if Y.dtype != torch.float32 or Y.stride(-1) != 1 or not Y.is_cuda:
        raise ValueError("spatial_accum: Y must be float32 CUDA with unit inner stride")
    if C.dtype != torch.float32 or C.stride(-1) != 1 or not C.is_cuda:
        raise ValueError("spatial_accum: C must be float32 CUDA with unit inner stride")
    dev = Y.device
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    tt = _i32(times, dev) if times is not None else None
    T = fid.numel() if fid is not None else (tt.numel() if tt is not None else Y.shape[0])
    P, K = Y.shape[1], C.shape[0]
    if A1 is None:
        A1 = torch.empty((P, K), dtype=torch.float32, device=dev)
        Cs = torch.empty((K, K), dtype=torch.float32, device=dev)
        accumulate = False
    Ct = C.t().contiguous()   # frame-major copy of the traces (K T floats): the kernel's matrix operands in 64-byte runs
    with _timed("spatial_accum"):
        rc = _lib.load().dnmf_spatial_accum(Y.data_ptr(), Y.stride(0), _ptr(fid), C.data_ptr(), C.stride(0), Ct.data_ptr(),
                                            Ct.stride(0), _ptr(tt), T, P, K, A1.data_ptr(), Cs.data_ptr(),
                                            int(bool(accumulate)), _stream())
    _lib.check(rc, "dnmf_spatial_accum")
    return A1, Cs


def mu_spatial(A, A1, Cs, D=None, gamma=0.0):
    """K6.  A (P,K) fp32 updated in place: A * A1 / (A Cs + gamma D + 1e-32)."""
    for t, n in ((A, "A"), (A1, "A1"), (Cs, "Cs")):
        _f32(t, n)
    if D is not None:
        _f32(D, "D")
    P, K = A.shape
    with _timed("mu_spatial"):
        rc = _lib.load().dnmf_mu_spatial(A.data_ptr(), A1.data_ptr(), Cs.data_ptr(), _ptr(D), float(gamma or 0.0), P, K,
                                         _stream())
    _lib.check(rc, "dnmf_mu_spatial")
    return A


def spatial_lists_setup(layout, K, sz):
    """Tile lists of the list-form footprint update from the boxes of ``pack_footprints_lists``: dict(tables int32,
    total) -- ``total`` floats of the compact A1 buffer, or -1 when a tile lists more than 32 neurons (use the dense K5)."""
    X, Y, Z = (int(s) for s in sz)
    lib = _lib.load()
    nt = lib.dnmf_spatial_lists_tiles(X, Y, Z)
    tables = torch.empty((2 * nt + 2 + 32 * nt,), dtype=torch.int32, device=layout["bbox"].device)
    _lib.check(lib.dnmf_spatial_lists_setup(layout["bbox"].data_ptr(), K, X, Y, Z, tables.data_ptr(), _stream()),
               "dnmf_spatial_lists_setup")
    return {"tables": tables, "total": int(tables[2 * nt].item()), "ntiles": nt}


def spatial_accum_lists(Y, C, sl, sz, K, frame_ids=None, times=None, A1c=None, Cs=None, workspace=None):
    """K5, list form.  Y (>=T, ldy) registered frames, C (K, ldc) -> A1c (total) compact sums, Cs (K,K)."""
    X, Yd, Z = (int(s) for s in sz)
    if Y.dtype != torch.float32 or Y.stride(-1) != 1 or not Y.is_cuda:
        raise ValueError("spatial_accum_lists: Y must be float32 CUDA with unit inner stride")
    if C.dtype != torch.float32 or C.stride(-1) != 1 or not C.is_cuda:
        raise ValueError("spatial_accum_lists: C must be float32 CUDA with unit inner stride")
    dev = Y.device
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    tt = _i32(times, dev) if times is not None else None
    T = fid.numel() if fid is not None else (tt.numel() if tt is not None else Y.shape[0])
    total = sl["total"]
    if A1c is None:
        A1c = torch.empty((total,), dtype=torch.float32, device=dev)
        Cs = torch.empty((K, K), dtype=torch.float32, device=dev)
    lib = _lib.load()
    need = lib.dnmf_spatial_accum_lists_workspace(X, Yd, Z, total, T)
    if need and (workspace is None or workspace.numel() * workspace.element_size() < need):
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    with _timed("spatial_accum_lists"):
        rc = lib.dnmf_spatial_accum_lists(Y.data_ptr(), Y.stride(0), _ptr(fid), C.data_ptr(), C.stride(0), _ptr(tt), T, X, Yd, Z, K,
                                          sl["tables"].data_ptr(), total, A1c.data_ptr(), Cs.data_ptr(), _ptr(workspace),
                                          0 if workspace is None else workspace.numel() * workspace.element_size(), _stream())
    _lib.check(rc, "dnmf_spatial_accum_lists")
    return A1c, Cs, workspace


def mu_spatial_lists(A, layout, sl, A1c, Cs, sz, D=None, gamma=0.0):
    """K6, list form: A (P,K) updated in place at the entries the tiles list; A1c is overwritten with the new values."""
    X, Y, Z = (int(s) for s in sz)
    for t, n in ((A, "A"), (A1c, "A1c"), (Cs, "Cs")):
        _f32(t, n)
    if D is not None:
        _f32(D, "D")
    with _timed("mu_spatial_lists"):
        rc = _lib.load().dnmf_mu_spatial_lists(A.data_ptr(), layout["At"].data_ptr(), A1c.data_ptr(), Cs.data_ptr(), _ptr(D),
                                               float(gamma or 0.0), X, Y, Z, A.shape[1], sl["tables"].data_ptr(), _stream())
    _lib.check(rc, "dnmf_mu_spatial_lists")
    return A


def pack_footprints_lists(A, sz):
    """Layout of K3n: dict(At (K,P), bbox (K,6) int32, pair_slot (K,K) int32, nslot int, boxfrac) -- ``boxfrac`` is
    the summed volume of the footprint boxes over the volume, i.e. the mean number of listed neurons per voxel."""
    X, Y, Z = (int(s) for s in sz)
    K = A.shape[-1]
    A2 = _f32(A.reshape(-1, K), "A")
    dev = A.device
    lib = _lib.load()
    At = torch.empty((K, halo_voxels(sz)), dtype=torch.float32, device=dev)   # halo layout, border zeroed by the call
    bbox = torch.empty((K, 6), dtype=torch.int32, device=dev)
    pair_slot = torch.empty((K, K), dtype=torch.int32, device=dev)
    nslot = torch.zeros((1,), dtype=torch.int32, device=dev)
    axis_masks = torch.empty((lib.dnmf_lists_axis_masks_bytes(X, Y, Z, K) // 8,), dtype=torch.int64, device=dev)
    _lib.check(lib.dnmf_pack_footprints_lists(A2.data_ptr(), X, Y, Z, K, At.data_ptr(), bbox.data_ptr(),
                                              pair_slot.data_ptr(), nslot.data_ptr(), axis_masks.data_ptr(), _stream()),
               "dnmf_pack_footprints_lists")
    bb = bbox.cpu().long()
    ext = (bb[:, 1::2] - bb[:, 0::2] + 1).clamp_min(0)
    ns = int(nslot.item())
    # per row of G the columns inside the pattern, ascending, padded with columns outside it (for K4)
    pattern = pair_slot != ns - 1
    widest = int(pattern.sum(1).max().item())
    NN = next((n for n in (8, 16, 32) if widest <= n <= K), None)
    nbr = None
    if NN is not None:
        nbr = torch.argsort((~pattern).to(torch.uint8), dim=1, stable=True)[:, :NN].to(torch.int32).contiguous()
    return {"At": At, "bbox": bbox, "pair_slot": pair_slot, "nslot": ns, "nbr": nbr, "axis_masks": axis_masks,
            "boxfrac": float(ext.prod(1).sum()) / (X * Y * Z)}


LISTS_MAX_SLOTS = 3800
# executed-work counters of the K3n launches while bench.py's TIMING is on (int64[2] tensor)
LISTS_COUNTERS = None


def warp_gram_rhs_lists(layout, K, sz, beta, times, frames, frame_ids=None, workspace=None, finish=True):
    """K3n.  Returns G (B,K,K), r (B,K), workspace; with ``finish=False`` G and r are None and the slot tables stay
    in ``workspace`` for ``mu_temporal_slots``."""
    global LISTS_COUNTERS
    X, Y, Z = (int(s) for s in sz)
    dev = beta.device
    _f32(beta, "beta")
    lib = _lib.load()
    tt = _i32(times, dev) if times is not None else None
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = tt.numel() if tt is not None else (fid.numel() if fid is not None else frames.shape[0])
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("warp_gram_rhs_lists: frames must be float32 CUDA with unit inner stride")
    nslot = layout["nslot"]
    need = lib.dnmf_warp_gram_rhs_lists_workspace(nslot, K, X, Y, Z, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    G = torch.empty((B, K, K), dtype=torch.float32, device=dev) if finish else None
    r = torch.empty((B, K), dtype=torch.float32, device=dev) if finish else None
    counters = None
    if TIMING is not None:
        if LISTS_COUNTERS is None:
            LISTS_COUNTERS = torch.zeros(12, dtype=torch.int64, device=dev)   # [2:] only in stamp builds
        counters = LISTS_COUNTERS
    with _timed("warp_gram_rhs_lists"):
        rc = lib.dnmf_warp_gram_rhs_lists(
            layout["At"].data_ptr(), layout["bbox"].data_ptr(), layout["pair_slot"].data_ptr(),
            layout["axis_masks"].data_ptr(), nslot, K, X, Y, Z,
            beta.data_ptr(), beta.shape[2], _ptr(tt), B, frames.data_ptr(), frames.stride(0), _ptr(fid), _ptr(G),
            _ptr(r), workspace.data_ptr(), workspace.numel() * workspace.element_size(), _ptr(counters), _stream())
    _lib.check(rc, "dnmf_warp_gram_rhs_lists")
    return G, r, workspace


def mu_temporal_slots(layout, workspace, sz, C, iters: int):
    """K4 straight from the slot tables ``warp_gram_rhs_lists(..., finish=False)`` left in ``workspace``; C (K,T) fp32
    updated in place (T = the frames of that launch).  Needs ``layout["nbr"]``."""
    X, Y, Z = (int(s) for s in sz)
    if not (C.is_cuda and C.dtype == torch.float32 and C.stride(1) == 1):
        raise ValueError("mu_temporal_slots: C must be float32 CUDA with unit inner stride")
    K, T = C.shape
    lib = _lib.load()
    nbr = layout["nbr"]
    with _timed("mu_temporal_slots"):
        rc = lib.dnmf_mu_temporal_slots(workspace.data_ptr(), lib.dnmf_warp_gram_rhs_lists_chunks(X, Y, Z, T),
                                        layout["nslot"], layout["pair_slot"].data_ptr(), C.data_ptr(), C.stride(0), K, T,
                                        int(iters), nbr.data_ptr(), nbr.shape[1], _stream())
    _lib.check(rc, "dnmf_mu_temporal_slots")
    return C


class Communicator:
    """C1: the library's RCCL communicator over the ranks of a ``torch.distributed`` group (one process per GPU).

    ``torch.distributed`` is only the out-of-band channel for the 128-byte id; the all-reduce itself is
    ``dnmf_allreduce_sum_f32`` on torch's current stream.  ``group=None`` with no initialised default group gives a
    one-rank communicator."""

    def __init__(self, group=None):
        import ctypes
        import torch.distributed as dist
        lib = _lib.load()
        multi = dist.is_available() and dist.is_initialized()
        self.nranks = dist.get_world_size(group) if multi else 1
        self.rank = dist.get_rank(group) if multi else 0
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.dnmf_comm_unique_id(ident), "dnmf_comm_unique_id")
        if self.nranks > 1:
            on_gpu = dist.get_backend(group) == "nccl"
            t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
            t = t.cuda() if on_gpu else t
            dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ctypes.create_string_buffer(bytes(t.cpu().tolist()), 128)
        handle = ctypes.c_void_p()
        _lib.check(lib.dnmf_comm_init(ctypes.byref(handle), ident, self.nranks, self.rank), "dnmf_comm_init")
        self._handle = handle

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over ranks of a contiguous fp32 CUDA tensor."""
        t = _f32(t, "all_reduce_")
        with _timed("allreduce"):
            rc = _lib.load().dnmf_allreduce_sum_f32(self._handle, _ptr(t), t.numel(), _stream())
        _lib.check(rc, "dnmf_allreduce_sum_f32")
        return t

    def close(self):
        if self._handle:
            _lib.check(_lib.load().dnmf_comm_destroy(self._handle), "dnmf_comm_destroy")
            self._handle = None


def image_iwarp(frames, frame_ids, sz, beta, times, out=None, exhaustive=False, count=None):
    """K7.  Registered frames (B,P): nearest-neighbour inverse warp under beta[:, :, times].  ``exhaustive``: search
    all P candidates for every lattice point (the checker of the window search); ``count``: int64[1] CUDA tensor
    incremented by the lattice points that needed the exhaustive search."""
    X, Y, Z = (int(s) for s in sz)
    P = X * Y * Z
    _f32(beta, "beta")
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("image_iwarp: frames must be float32 CUDA with unit inner stride")
    dev = frames.device
    tt = _i32(times, dev)
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = tt.numel()
    if out is None:
        out = torch.empty((B, P), dtype=torch.float32, device=dev)
    if out.shape[0] < B or out.stride(0) < P or out.stride(1) != 1:
        raise ValueError("image_iwarp: out must be (>=B, ld) with ld >= P")
    lib = _lib.load()
    # frames per launch (gridDim.y); one flag byte per lattice point of the launch: at most 256 MiB of flags (a 512x512x20
    # volume would otherwise ask for 86 GB per 16384-frame launch)
    step = max(1, min(16384, (256 << 20) // P))
    ws = torch.empty((lib.dnmf_image_iwarp_workspace(X, Y, Z, min(B, step)),), dtype=torch.uint8, device=dev)
    for s in range(0, B, step):
        n = min(step, B - s)
        src = frames if fid is not None else frames[s:]   # without ids, frame b of a launch is its row b
        with _timed("image_iwarp"):
            rc = (lib.dnmf_image_iwarp(src.data_ptr(), frames.stride(0), 0 if fid is None else fid[s:].data_ptr(), X, Y, Z,
                                        beta.data_ptr(), beta.shape[2], tt[s:].data_ptr(), n, out[s:].data_ptr(),
                                        out.stride(0), ws.data_ptr(), ws.numel(), int(bool(exhaustive)), _ptr(count),
                                        _stream()))
        _lib.check(rc, "dnmf_image_iwarp")
    return out


def patch_grid(sz, strides, overlaps):
    """The patch grid of the reference's ``sliding_window_3d`` (MotionCorrect.py:1190-1221): ``(dims (3,), starts (NP,3))`` as
    numpy int arrays, patches in the reference's order (x outermost)."""
    import ctypes

    import numpy as np
    X, Y, Z = (int(s) for s in sz)
    st = (ctypes.c_int * 3)(*[int(v) for v in strides])
    ov = (ctypes.c_int * 3)(*[int(v) for v in overlaps])
    dims = (ctypes.c_int * 3)()
    lib = _lib.load()
    NP = lib.dnmf_register_patches_grid(X, Y, Z, st, ov, dims, None)
    if NP <= 0:
        raise ValueError(f"patch_grid: windows of strides {tuple(strides)} + overlaps {tuple(overlaps)} do not fit the volume {X}x{Y}x{Z}")
    starts = (ctypes.c_int * (3 * NP))()
    lib.dnmf_register_patches_grid(X, Y, Z, st, ov, dims, starts)
    return np.array(list(dims)), np.array(list(starts)).reshape(NP, 3)


def register_patches(frames, tmpl, sz, strides, overlaps, max_shifts, max_deviation_rigid=3, upsample_factor=10,
                     add_to_movie=0.0, frame_ids=None):
    """K8.  frames (>=B, P) fp32 CUDA rows, tmpl (P) -> rigid shifts (B,3) and per-patch shifts (B,NP,3) (signs -x, -y, +z:
    the reference's ``x/y/z_shifts_els``)."""
    import ctypes
    X, Y, Z = (int(s) for s in sz)
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("register_patches: frames must be float32 CUDA with unit inner stride")
    dev = frames.device
    tm = _f32(tmpl.reshape(-1), "tmpl")
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = fid.numel() if fid is not None else frames.shape[0]
    st = (ctypes.c_int * 3)(*[int(v) for v in strides])
    ov = (ctypes.c_int * 3)(*[int(v) for v in overlaps])
    ms = (ctypes.c_int * 3)(*[int(v) for v in max_shifts])
    lib = _lib.load()
    NP = lib.dnmf_register_patches_grid(X, Y, Z, st, ov, None, None)
    if NP <= 0:
        raise ValueError(f"register_patches: windows of strides {tuple(strides)} + overlaps {tuple(overlaps)} do not fit the volume")
    need = lib.dnmf_register_patches_workspace(X, Y, Z, st, ov, B)
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    rigid = torch.empty((B, 3), dtype=torch.float32, device=dev)
    patch = torch.empty((B, NP, 3), dtype=torch.float32, device=dev)
    with _timed("register_patches"):
        rc = lib.dnmf_register_patches(frames.data_ptr(), frames.stride(0), _ptr(fid), B, tm.data_ptr(), X, Y, Z, st, ov, ms,
                                       int(max_deviation_rigid), int(upsample_factor), float(add_to_movie), rigid.data_ptr(),
                                       patch.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "dnmf_register_patches")
    return rigid, patch


_BORDER = {False: 0, True: 1, 'min': 2, 'copy': 3}     # border_nan of apply_shifts_dft (MotionCorrect.py:1098-1145)


def rigid_correct(frames, tmpl, sz, max_shifts, upsample_factor=10, add_to_movie=0.0, border_nan=True, frame_ids=None,
                  want_frames=False, tsum=None, tcount=None):
    """K8, rigid pass.  frames (>=B, P) fp32 CUDA rows, tmpl (P) -> (rigid shifts (B,3) as register_translation_3d returns
    them, corrected frames (B,P) or None, tsum (P) fp32, tcount (P) int32): the per-voxel sums and counts of the finite
    corrected values are ADDED into ``tsum`` / ``tcount`` when given (a video walked in pieces), else start from zero."""
    import ctypes
    X, Y, Z = (int(s) for s in sz)
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("rigid_correct: frames must be float32 CUDA with unit inner stride")
    dev = frames.device
    P = X * Y * Z
    tm = _f32(tmpl.reshape(-1), "tmpl")
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = fid.numel() if fid is not None else frames.shape[0]
    ms = (ctypes.c_int * 3)(*[int(v) for v in max_shifts])
    lib = _lib.load()
    ws = torch.empty((lib.dnmf_rigid_correct_workspace(X, Y, Z, B),), dtype=torch.uint8, device=dev)
    rigid = torch.empty((B, 3), dtype=torch.float32, device=dev)
    out = torch.empty((B, P), dtype=torch.float32, device=dev) if want_frames else None
    if tsum is None:
        tsum, tcount = torch.zeros(P, dtype=torch.float32, device=dev), torch.zeros(P, dtype=torch.int32, device=dev)
    with _timed("rigid_correct"):
        rc = lib.dnmf_rigid_correct(frames.data_ptr(), frames.stride(0), _ptr(fid), B, tm.data_ptr(), X, Y, Z, ms,
                                    int(upsample_factor), float(add_to_movie), _BORDER[border_nan], rigid.data_ptr(),
                                    _ptr(out), P, tsum.data_ptr(), tcount.data_ptr(), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "dnmf_rigid_correct")
    return rigid, out, tsum, tcount


def apply_shifts_points(points, patch_shifts, centers):
    """points (K,3), patch_shifts (T,NP,3), centers (NP,3) -> (K,3,T) fp32 (MotionCorrect.apply_shifts_points)."""
    pts, sh, ce = _f32(points, "points"), _f32(patch_shifts, "patch_shifts"), _f32(centers, "centers")
    K, (T, NP, _) = pts.shape[0], sh.shape
    out = torch.empty((K, 3, T), dtype=torch.float32, device=pts.device)
    _lib.check(_lib.load().dnmf_apply_shifts_points(pts.data_ptr(), K, sh.data_ptr(), T, NP, ce.data_ptr(), out.data_ptr(), _stream()),
               "dnmf_apply_shifts_points")
    return out


def pack_footprints_sparse(A, order):
    """A (..., K) and a neuron order -> (Aps (P,Ks), row_mask (P) uint8) for the zero-skipping Gram kernel."""
    K = A.shape[-1]
    A2 = _f32(A.reshape(-1, K), "A")
    lib = _lib.load()
    Ks = lib.dnmf_sparse_k(K)
    od = _i32(order, A.device)
    Aps = torch.empty((A2.shape[0], Ks), dtype=torch.float32, device=A.device)
    mask = torch.empty((A2.shape[0],), dtype=torch.uint8, device=A.device)
    _lib.check(lib.dnmf_pack_footprints_sparse(A2.data_ptr(), A2.shape[0], K, od.data_ptr(), Aps.data_ptr(), Ks,
                                               mask.data_ptr(), _stream()), "dnmf_pack_footprints_sparse")
    return Aps, mask


# when bench.py sets TIMING it also finds here the executed-work counters of the K3s launches (int64[2] tensor)
SPARSE_COUNTERS = None


# which zero-skipping kernel warp_gram_rhs_sparse runs: 'table' (local block table, 4 waves per SIMD) or 'static'
SPARSE_VARIANT = 'table'


def warp_gram_rhs_sparse(Aps, K, order, row_mask, sz, beta, times, frames, frame_ids=None, workspace=None, variant=None):
    """K3s.  Returns G (B,K,K), r (B,K) in the original neuron order."""
    global SPARSE_COUNTERS
    lt = (variant or SPARSE_VARIANT) == 'table'
    X, Y, Z = (int(s) for s in sz)
    P = X * Y * Z
    dev = Aps.device
    _f32(Aps, "Aps"), _f32(beta, "beta")
    lib = _lib.load()
    od = _i32(order, dev)
    tt = _i32(times, dev) if times is not None else None
    fid = _i32(frame_ids, dev) if frame_ids is not None else None
    B = tt.numel() if tt is not None else (fid.numel() if fid is not None else frames.shape[0])
    if frames.dtype != torch.float32 or frames.stride(-1) != 1 or not frames.is_cuda:
        raise ValueError("warp_gram_rhs_sparse: frames must be float32 CUDA with unit inner stride")
    need = (lib.dnmf_warp_gram_rhs_sparse_lt_workspace if lt else lib.dnmf_warp_gram_rhs_sparse_workspace)(P, K, B)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(((need + 3) // 4,), dtype=torch.float32, device=dev)
    G = torch.empty((B, K, K), dtype=torch.float32, device=dev)
    r = torch.empty((B, K), dtype=torch.float32, device=dev)
    counters = None
    if TIMING is not None:
        if SPARSE_COUNTERS is None:
            SPARSE_COUNTERS = torch.zeros(2, dtype=torch.int64, device=dev)
        counters = SPARSE_COUNTERS
    with _timed("warp_gram_rhs_sparse"):
        rc = (lib.dnmf_warp_gram_rhs_sparse_lt if lt else lib.dnmf_warp_gram_rhs_sparse)(
            Aps.data_ptr(), Aps.shape[-1], K, od.data_ptr(), row_mask.data_ptr(), X, Y, Z, beta.data_ptr(),
            beta.shape[2], _ptr(tt), B, frames.data_ptr(), frames.stride(0), _ptr(fid), G.data_ptr(), r.data_ptr(),
            workspace.data_ptr(), workspace.numel() * workspace.element_size(), _ptr(counters), _stream())
    _lib.check(rc, "dnmf_warp_gram_rhs_sparse")
    return G, r, workspace
