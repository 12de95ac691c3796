"""Deformable NMF demixing on MI355X: the Python surface of the reference's ``Demix/dNMF.py``
(``ExponentialFP``, ``DeformableNMF``, ``SimulatedVideoDataset``) on hand-written HIP kernels.

The classes keep the reference's constructor arguments, attributes and method signatures so that its
``demo.py`` runs against this module unchanged (``from Demix.dNMF import ...`` resolves to this file
through the ``Demix`` package at the repository root).  What is different is underneath:

* ``ExponentialFP.forward`` (reference ``Demix/dNMF.py:53-62``) calls K1 ``dnmf_warp_gather`` for the
  materialised ``A_t`` / ``grid`` and the fused path (``dnmf_recon_image`` + K2 ``dnmf_warp_recon_grad``)
  for ``A_tC``; ``A_tC`` carries an autograd node whose backward is K2 again, so
  ``F.mse_loss(A_tC, batch).backward()`` deposits ``beta.grad`` like the reference.
* ``DeformableNMF.update_motion`` (``:181-194``) runs one fused K2 launch per mini-batch and hands
  ``beta.grad`` to the caller's optimiser; it never materialises the K warped footprints.
* ``DeformableNMF.update_footprints`` (``:163-179``) computes the per-frame Gram matrices and right-hand
  sides once with K3 ``dnmf_warp_gram_rhs`` (fp32 MFMA) and iterates the multiplicative update with K4
  ``dnmf_mu_temporal``; the reference recomputes both contractions in every iteration from a float64
  ``A_t`` of all frames.

There is no CPU fallback: without ``libdnmf_hip.so`` and a GPU these classes raise.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset

from .. import ops, sharding
from ..WUtils import Simulator

device = 'cuda'

# update_footprints returns the reference's dense (X,Y,Z,K,T) float64 A_t only below this many bytes
DENSE_RETURN_LIMIT = 1 << 31
# update_motion keeps the reconstruction images of all T frames resident below this many bytes
LISTS_BOXFRAC_LIMIT = 6.0   # 'auto' takes K3n below this mean number of footprint boxes per voxel
RECON_CACHE_LIMIT = 64 << 30
K2_MAX_FRAMES = 32768       # frames per K2 launch (they ride on gridDim.y)
STAGE_LIMIT = 96 << 30      # a host loader's frames are staged on the GPU once per pass below this many bytes


def _sz_list(sz):
    return [int(s) for s in (sz.tolist() if isinstance(sz, torch.Tensor) else sz)]


class _WarpRecon(torch.autograd.Function):
    """A_tC = trilinear warp of the reconstruction image S = A.C_t; differentiable w.r.t. beta."""

    @staticmethod
    def forward(ctx, beta, fp, times, C):
        times_t = torch.as_tensor(list(times), dtype=torch.int32, device=beta.device)
        Csel = C.to(device=beta.device, dtype=torch.float32).contiguous()
        S = fp.recon_image(Csel, times_t)
        out = ops.warp_recon_grad(S, None, None, None, fp.sz_list, beta.detach(), times_t, grad=None,
                                  gout=torch.zeros((len(times_t), fp.P), dtype=torch.float32, device=beta.device),
                                  want_recon=True, want_loss=False, want_reg=True)
        ctx.fp, ctx.S, ctx.times_t = fp, S, times_t
        ctx.save_for_backward(beta)
        ctx.mark_non_differentiable(out["reg"])
        return out["recon"].view(len(times_t), *fp.sz_list), out["reg"]

    @staticmethod
    def backward(ctx, g_recon, _g_reg):
        (beta,) = ctx.saved_tensors
        grad = torch.zeros_like(beta)
        gout = g_recon.contiguous().view(len(ctx.times_t), ctx.fp.P).float()
        ops.warp_recon_grad(ctx.S, None, None, None, ctx.fp.sz_list, beta.detach(), ctx.times_t, grad=grad, gout=gout,
                            want_recon=False, want_loss=False, want_reg=False)
        return grad, None, None, None


class ExponentialFP(nn.Module):
    """Gaussian footprints ``A`` deformed per frame by a quadratic map with coefficients ``beta``.

    Mirrors reference ``Demix/dNMF.py:18-122``: same constructor, same attributes (``beta`` leaf
    (10,3,T) with requires_grad, ``A`` (X,Y,Z,K), ``sigma``, ``pos``, ``sz``, ``flow_id``,
    ``transformed``), same ``forward(times, C) -> (A_tC, A_t, grid, reg)``.
    """

    def __init__(self, sz, K, T, positions=None, shape_std=3):
        super().__init__()
        ops._lib.load()  # fail here, loudly, if the HIP library is not built
        sz_t = torch.as_tensor(sz)
        self.sz_list = _sz_list(sz_t)
        X, Y, Z = self.sz_list
        self.K, self.T, self.P = int(K), int(T), X * Y * Z

        # voxel lattice and its quadratic basis (reference :22-23); kept for attribute compatibility --
        # the kernels regenerate both from the voxel index
        gx, gy, gz = torch.meshgrid(torch.arange(X), torch.arange(Y), torch.arange(Z), indexing='ij')
        flow_id = torch.stack((gx, gy, gz), 3).float().to(device)
        self.flow_id = flow_id
        self.transformed = ExponentialFP.quadratic_basis(flow_id)

        beta = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].repeat(1, 1, T)
        self.beta = beta.to(device).contiguous()
        self.beta.requires_grad = True

        self.sigma = (torch.ones(K) * shape_std).to(device)
        if positions is None:
            self.pos = (1 + torch.rand(K, 3) * sz_t[None, :]).to(device)
        else:
            self.pos = positions.to(device)
        self.sz = sz_t.to(device)

        # A[x,y,z,k] = exp(-sum_d (coord_d - pos[k,d])^2 / sigma_k^2), neuron by neuron to bound the
        # temporary (the reference builds an (X,Y,Z,3,K) tensor at once, :39-40)
        A = torch.empty((X, Y, Z, K), dtype=torch.float32, device=device)
        for k in range(K):
            A[..., k] = torch.exp((-(flow_id - self.pos[k][None, None, None, :].float()) ** 2
                                   / self.sigma[k] ** 2).sum(3))
        self.A = A
        self._packed = None
        self._packed_version = None
        self._sparse = None
        self._sparse_version = None
        self._sparse_pairs = None
        self._sparse_pairs_version = None
        self._lists = None
        self._lists_version = None
        self.use_lists = True   # reconstruction image from neuron lists when the footprints are compact

    def invalidate_layouts(self):
        """Forget every packed copy of ``A``.  The packed copies are keyed on ``(A.data_ptr(), A._version)``, which a
        kernel that writes ``A`` through its raw pointer (K6) does not change: such callers say so here."""
        self._packed = self._sparse = self._sparse_pairs = self._lists = None

    @staticmethod
    def quadratic_basis(P):
        """[1, x, y, z, x^2, y^2, z^2, xy, xz, yz] (reference :46-51)."""
        x, y, z = P[..., 0:1], P[..., 1:2], P[..., 2:3]
        return torch.cat((x * 0 + 1, P, P * P, x * y, x * z, y * z), -1)

    def packed_footprints(self):
        """(P,Kp) zero-padded copy of ``A`` for the MFMA kernels; rebuilt when ``A`` is replaced or edited."""
        key = (self.A.data_ptr(), self.A._version)
        if self._packed is None or self._packed_version != key:
            self._packed = ops.pack_footprints(self.A.contiguous())
            self._packed_version = key
        return self._packed

    def packed_columns(self, cols):
        """Packed copy of the footprints of the neurons ``cols`` only (K > 127 is handled by column groups: the MFMA
        kernels hold at most 8 blocks of 16 channels in registers)."""
        sub = self.A.reshape(self.P, self.K)[:, torch.as_tensor(cols, device=self.A.device)].contiguous()
        return ops.pack_footprints(sub)

    def recon_image(self, C, times, out=None, zero_state=None):
        """S[b] = A . C[:, times[b]] as (B, >= ops.halo_voxels(sz)) rows in the halo layout K2 gathers from: from the
        neuron lists when the footprints are compact (``use_lists``), else ``dnmf_recon_image`` (fp32 MFMA), by groups
        of 112 neurons when K > 127."""
        if self.use_lists and self.K <= 256:
            ly = self.packed_lists()
            if ly["boxfrac"] < LISTS_BOXFRAC_LIMIT:
                # zero_state: a one-entry list the owner of a persistent ``out`` keeps -- the layout whose empty tiles are
                # known to hold zeros in these rows (written by a call without skipping); None after any other writer
                skip = zero_state is not None and zero_state[0] is ly
                res = ops.recon_image_lists(ly, self.K, self.sz_list, C, times, out=out, skip_empty=skip)
                if zero_state is not None:
                    zero_state[0] = ly
                return res
        if zero_state is not None:
            zero_state[0] = None
        if self.K <= 127:
            return ops.recon_image(self.packed_footprints(), self.K, self.sz_list, C, times, out=out)
        for n, s0 in enumerate(range(0, self.K, 112)):
            cols = list(range(s0, min(self.K, s0 + 112)))
            part = ops.recon_image(self.packed_columns(cols), len(cols), self.sz_list, C[cols].contiguous(), times,
                                   out=out if n == 0 else None)
            if n == 0:
                out = part
            else:
                out[:part.shape[0], :part.shape[1]] += part
        return out

    def _zorder(self):
        """Neuron indices sorted along a Z-order curve of the footprint centroids (x,y), int32 on the GPU."""
        A2 = self.A.reshape(self.P, self.K)
        mass = A2.sum(0).clamp_min(1e-30)
        cen = (self.flow_id.reshape(self.P, 3).T @ A2) / mass            # (3,K) centroids
        q = (cen[:2] / torch.tensor(self.sz_list[:2], device=cen.device)[:, None]).clamp(0, 1 - 1e-6)
        q = (q * 1024).long().cpu().numpy()
        code = np.zeros(self.K, dtype=np.int64)
        for bit in range(10):                                           # interleave the bits of x and y
            code |= ((q[0] >> bit) & 1) << (2 * bit + 1) | ((q[1] >> bit) & 1) << (2 * bit)
        return torch.from_numpy(np.argsort(code, kind="stable").astype(np.int32)).to(device)

    @staticmethod
    def _sparse_layout(A2, order):
        Aps, mask = ops.pack_footprints_sparse(A2, order)
        nb = Aps.shape[1] // 16
        bits = (mask[:, None] >> torch.arange(nb, device=mask.device, dtype=torch.uint8)[None, :]) & 1
        return {"Aps": Aps, "order": order, "row_mask": mask, "occupancy": float(bits.float().mean())}

    def packed_sparse(self):
        """Layout for the zero-skipping Gram kernel K3s: neurons ordered along a Z-order curve of their footprint
        centroids, ``Aps`` (P,Ks) in that order, one block-occupancy byte per footprint row, and the mean
        fraction of 16-neuron blocks that are non-zero per row (``occupancy``).  None if K > 128."""
        if self.K > 128:
            return None
        key = (self.A.data_ptr(), self.A._version)
        if self._sparse is None or self._sparse_version != key:
            self._sparse = self._sparse_layout(self.A.contiguous(), self._zorder())
            self._sparse_version = key
        return self._sparse

    # Footprint values below ``footprint_floor`` are left out of the neuron lists (K3n, the list reconstruction, the list
    # form of the footprint update): 0.0 -- the default -- keeps every non-zero value.  A Gaussian footprint exp(-d^2 / 9)
    # is a non-zero fp32 number out to 30 voxels (1e-45), so its box is 61 x 61; the values beyond ~15 voxels (1e-10) enter
    # sums of order 1-10 and cannot change an fp32 result except through the ORDER of the summation (measured: the Gram
    # data and the traces move by 4e-7 relative, as they do between K3n's own launch forms).  An extension: the reference
    # has no such knob.  bench.py reports it as ``extras.footprint_floor``; the headline runs with 0.0.
    footprint_floor = 0.0

    def packed_lists(self):
        """Layout of the neuron-list Gram kernel K3n (``ops.pack_footprints_lists``), rebuilt when ``A`` changes."""
        key = (self.A.data_ptr(), self.A._version, float(self.footprint_floor))
        if self._lists is None or self._lists_version != key:
            A = self.A.contiguous()
            if self.footprint_floor > 0:
                A = torch.where(A < self.footprint_floor, torch.zeros((), dtype=A.dtype, device=A.device), A)
            self._lists = ops.pack_footprints_lists(A, self.sz_list)
            self._lists_version = key
        return self._lists

    def packed_sparse_pairs(self, group=64):
        """K > 128: the neurons, in Z-order, are cut into groups of ``group`` and every pair of groups gets its own
        K3s layout (a mask byte holds 8 blocks of 16).  Returns ``[(cols (n,) long, layout), ...]``; ``cols`` are
        the original neuron indices of the pair, in the column order of ``layout["Aps"]`` before its own sort."""
        key = (self.A.data_ptr(), self.A._version, group)
        if self._sparse_pairs is None or self._sparse_pairs_version != key:
            z = self._zorder().long()
            groups = [z[s0:s0 + group] for s0 in range(0, self.K, group)]
            A2 = self.A.reshape(self.P, self.K)
            pairs = []
            for i in range(len(groups)):
                for j in range(i + 1, len(groups)):
                    cols = torch.cat([groups[i], groups[j]])
                    sub = A2[:, cols].contiguous()
                    ident = torch.arange(cols.numel(), dtype=torch.int32, device=device)  # already Z-ordered
                    pairs.append((cols, self._sparse_layout(sub, ident)))
            self._sparse_pairs, self._sparse_pairs_version = pairs, key
        return self._sparse_pairs

    def forward(self, times, C):
        """Returns ``(A_tC (B,X,Y,Z), A_t (B,K,X,Y,Z), grid (X,Y,Z,3,B), reg (B))`` for the frames ``times``."""
        times = [int(t) for t in (times.tolist() if hasattr(times, 'tolist') else times)]
        A_tC, reg = _WarpRecon.apply(self.beta, self, times, C)
        A_t, grid = ops.warp_gather(self.A.contiguous(), self.beta.detach(), times)
        return A_tC, A_t, grid, reg

    @staticmethod
    def log_det_jac(B, P):
        """log|det J| of the quadratic map at ``P`` (reference :107-122, with its 8/9 row convention)."""
        x, y, z = P[0], P[1], P[2]
        # column c of the Jacobian as the reference writes it: rows 8 / 9 of beta act as the yz / xz terms
        col = [(B[1, c] + 2 * B[4, c] * x + B[7, c] * y + B[9, c] * z,
                B[2, c] + 2 * B[5, c] * y + B[7, c] * x + B[8, c] * z,
                B[3, c] + 2 * B[6, c] * z + B[8, c] * y + B[9, c] * x) for c in range(3)]
        (a, b, c), (d, e, f), (g, h, i) = col
        return torch.log(abs(a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)))

    @staticmethod
    def spatial_pushforward(dl, batch_size, sz, device, model):
        """Reference :69-93: ``A_t`` (X,Y,Z,K,T), the registered video ``Y_i`` and the raw video ``Y`` (X,Y,Z,T)
        as float64 numpy for every frame the loader yields (K1 + K7)."""
        X, Y_, Z = _sz_list(sz)
        A_list, Y_list, Yi_list = [], [], []
        for data in dl:
            times = data[1].tolist()
            beta = model.fp.beta.detach()
            A_t, _ = ops.warp_gather(model.fp.A.contiguous(), beta, times, want_grid=False)
            A_list.append(A_t.permute(2, 3, 4, 1, 0).double().cpu().numpy())
            Y_list.append(data[0].permute(1, 2, 3, 0).double().cpu().numpy())
            fr = data[0].to(device, torch.float32).reshape(len(times), -1).contiguous()
            Yi = ops.image_iwarp(fr, None, (X, Y_, Z), beta, times)
            Yi_list.append(Yi.view(len(times), X, Y_, Z).permute(1, 2, 3, 0).double().cpu().numpy())
        return np.concatenate(A_list, 4), np.concatenate(Yi_list, 3), np.concatenate(Y_list, 3)


class DeformableNMF:
    """Reference ``Demix/dNMF.py:124-194``: owns the spatial model ``fp`` and the traces ``C`` (K,T)."""

    def __init__(self, sz, K, T, positions=None):
        self.SpatialModel = ExponentialFP
        self.fp = self.SpatialModel(sz=sz, K=K, T=T, positions=positions)
        self.C = torch.rand((K, T)).to(device)
        self.A = torch.rand((K, _sz_list(sz)[0], _sz_list(sz)[1])).to(device)  # unused in the reference too (:131)
        self.verbose = True
        if positions is not None:
            # D = 1 - exp(-0.01 * distance(voxel, centre_k)), float64 (reference :133-137)
            lat = self.fp.flow_id.reshape(-1, 3).double()
            d = torch.cdist(lat, positions.to(device).double())
            X, Y, Z = _sz_list(sz)
            self.D = (1 - torch.exp(-.01 * d)).reshape(X, Y, Z, K).cpu().numpy()
        else:
            self.D = None
        self._ws_k2 = None
        self._ws_k3 = None
        self._S_bufs = None
        self._S_zero = None
        self._gram_nbr = None   # (K,NN) columns of the last Gram matrices that can be non-zero, when K3n made them
        # update_motion evaluates a whole epoch per launch when the caller's optimiser is a plain
        # torch.optim.Adam on [fp.beta] and the loader is a ResidentLoader (same result, see _motion_epoch)
        self.fused_motion = True
        # Gram kernel: 'dense' = K3 (every product evaluated), 'sparse' = K3s (16-neuron blocks that are exactly zero
        # skipped), 'lists' = K3n (only the neurons whose non-zero box a tile of voxels can reach; same sums),
        # 'auto' = K3n while a voxel lies in few boxes, else K3s when on average fewer than half of the 16-neuron
        # blocks of a footprint row are non-zero, else K3, 'bf16' = K3b (every product, operands rounded to bf16: reduced precision the
        # reference does not have, never chosen by 'auto')
        self.gram_kernel = 'auto'
        # footprint update (spatial_step): 'lists' = K5 / K6 on (tile, listed neuron) pairs only, 'dense' = the MFMA K5 on all
        # P x K entries, 'auto' = lists while the footprints are compact
        self.spatial_kernel = 'auto'
        self._sl = None
        self._ws_k5 = None
        # torch.distributed group when this object holds one contiguous T-shard per rank (rank order = frame order);
        # used where the path has a real exchange: the neighbour term of update_temporal and spatial_step
        self.group = None
        self._comm = None  # ops.Communicator over self.group, built by spatial_step when collective == "c1"
        # the one collective of the path (spatial_step): 'torch' = torch.distributed.all_reduce on self.group (RCCL under the
        # "nccl" backend); 'c1' = dnmf_allreduce_sum_f32 on the library's own communicator.  C1 has only ever run with one
        # rank (one GPU per test box), so it is not the default
        self.collective = "torch"
        self._spatial_buf = None   # A1 (P,K) and C_s (K,K) of spatial_step, one buffer = one all-reduce
        self._ws_mg = None
        # > 0: the fused motion epoch on compact footprints makes its reconstruction images this many frames at a time
        # into one small buffer (dnmf_motion_grad_lists) instead of keeping the images of all T frames (4.5 GB at
        # 512x512x4000): same results bit for bit, same speed within 2 % from 128 frames on (the hoped-for gain from
        # images that stay in the Infinity Cache did not materialise), so it is a memory option; 0 = all frames at once
        self.motion_chunk = 0
        self._stage_buf = None     # device copy of the frames a host loader served in its last pass
        self._reg_buf = None       # registered frames (K7) of the last update_footprints(live_spatial=True)
        self._D_dev = None         # (id(self.D), fp32 device copy of D flattened to (P,K))
        self.stream_loader = True  # stage host loaders on the GPU once per pass (see _stage_epoch)
        self._warned = set()

    # ---- static NMF updates (numpy in / numpy out like the reference) ---------------------------------
    @staticmethod
    def update_temporal(A_t, C, Y, gamma=None):
        """One multiplicative update of ``C`` given explicit warped footprints (reference :139-149).
        ``A_t`` (X,Y,Z,K,T), ``C`` (K,T), ``Y`` (X,Y,Z,T) numpy; returns float64 numpy (K,T).  The contraction runs on
        the fp32 matrix cores (K3 without a warp: every voxel's own row of ``A_t``), the update in float64 (K4); more than
        127 neurons go by pairs of neuron groups (one K3 launch holds 127)."""
        A_t, C, Y = np.asarray(A_t), np.asarray(C), np.asarray(Y)
        X, Y_, Z, K, T = A_t.shape
        P = X * Y_ * Z
        dev = torch.device(device)
        A_dev = torch.from_numpy(np.ascontiguousarray(np.moveaxis(A_t, 4, 0).reshape(T, P, K))).to(dev, torch.float32)
        frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(Y, 3, 0).reshape(T, P))).to(dev, torch.float32)

        def gram(cols):
            sub = A_dev if cols is None else A_dev[:, :, cols].contiguous()
            Apk = ops.pack_footprints(sub)                                 # (T*P, Kp)
            return ops.warp_gram_rhs(Apk, sub.shape[2], (X, Y_, Z), None, list(range(T)), frames,
                                     a_frame_stride=P * Apk.shape[1])[:2]

        if K <= 127:
            G, r = gram(None)
        else:
            # one K3 launch holds 127 neurons: groups of 56, every PAIR of groups one launch on their union (both diagonal
            # blocks and the off-diagonal block of the pair), as DeformableNMF._gram_rhs_grouped does on the fit path
            G = torch.empty((T, K, K), dtype=torch.float32, device=dev)
            r = torch.empty((T, K), dtype=torch.float32, device=dev)
            groups = [list(range(s0, min(K, s0 + 56))) for s0 in range(0, K, 56)]
            for i in range(len(groups)):
                for j in range(i + 1, len(groups)):
                    idx = torch.as_tensor(groups[i] + groups[j], device=dev)
                    Gp, rp = gram(idx)
                    G[:, idx[:, None], idx[None, :]] = Gp
                    r[:, idx] = rp
        return _mu_temporal(G, r, torch.from_numpy(np.asarray(C, dtype=np.float64)).to(dev), gamma, 1).cpu().numpy()

    @staticmethod
    def update_spatial(A, C, Y_i, D=None, gamma=None):
        """One multiplicative update of un-warped footprints (reference :151-160): numpy in, float64 numpy out.
        ``A`` (..., K), ``C`` (K,T), ``Y_i`` (..., T), ``D`` None or like ``A``; the leading axes are the voxels
        (the reference's einsum strings accept two of them, any number works here).  K5 + K6, fp32 MFMA."""
        A, C, Y_i = np.asarray(A), np.asarray(C), np.asarray(Y_i)
        K, T = C.shape
        dev = torch.device(device)
        A_dev = torch.from_numpy(np.ascontiguousarray(A.reshape(-1, K))).to(dev, torch.float32)
        Y_dev = torch.from_numpy(np.ascontiguousarray(Y_i.reshape(-1, T).T)).to(dev, torch.float32)
        C_dev = torch.from_numpy(np.ascontiguousarray(C)).to(dev, torch.float32)
        D_dev = None if D is None else torch.from_numpy(np.ascontiguousarray(np.asarray(D).reshape(-1, K))).to(dev, torch.float32)
        if K <= 128:
            A1, Cs = ops.spatial_accum(Y_dev, C_dev)
        else:  # K5 holds 8 trace blocks per wave: columns of A1 by groups; C C^T is tiny
            A1 = torch.cat([ops.spatial_accum(Y_dev, C_dev[s0:s0 + 128].contiguous())[0] for s0 in range(0, K, 128)], 1)
            Cs = (C_dev.double() @ C_dev.double().T).float()
        ops.mu_spatial(A_dev, A1.contiguous(), Cs, D_dev, gamma)
        return A_dev.double().cpu().numpy().reshape(A.shape)

    def spatial_step(self, registered, D=None, gamma=None, frame_ids=None, times=None):
        """One multiplicative update of ``fp.A`` from the registered frames this process holds (the update the
        reference leaves commented out at :174, on the flattened voxel axis): K5 on the local frames, ONE all-reduce
        (sum) of the buffer that holds ``A1`` (P,K) and ``C_s`` (K,K) over ``self.group`` when the T axis is sharded,
        then K6 -- every rank ends with the same footprints.  ``registered`` (>=T_local,P) fp32 CUDA rows, frame b in
        row ``frame_ids[b]`` (None: b) with trace column ``times[b]`` (None: ``frame_ids[b]``, else b); ``D`` None or
        (X,Y,Z,K).  ``self.last_spatial_ms`` receives (K5, all-reduce, K6) HIP-event times when ``self.time_spatial``
        is set."""
        fp = self.fp
        P, K = fp.P, fp.K
        C = self.C.to(device, torch.float32).contiguous()
        if times is None:
            times = frame_ids
        sl = self._spatial_lists()
        # A1 and C_s live in one buffer so that the exchange is a single collective; with compact footprints A1 exists
        # only for (tile, listed neuron) pairs -- a few floats per voxel instead of K (3 MB instead of 105 MB at cfg 3)
        n1 = sl["total"] if sl is not None else P * K
        if self._spatial_buf is None or self._spatial_buf.numel() != n1 + K * K:
            self._spatial_buf = None
            self._spatial_buf = torch.empty((n1 + K * K,), dtype=torch.float32, device=device)
        buf = self._spatial_buf
        Cs = buf[n1:].view(K, K)
        if sl is not None:
            A1 = buf[:n1]
            _, _, self._ws_k5 = ops.spatial_accum_lists(registered, C, sl, fp.sz_list, K, frame_ids=frame_ids, times=times, A1c=A1,
                                                        Cs=Cs, workspace=getattr(self, "_ws_k5", None))
        else:
            A1 = buf[:n1].view(P, K)
            if K <= 128:
                ops.spatial_accum(registered, C, frame_ids=frame_ids, times=times, A1=A1, Cs=Cs, accumulate=False)
            else:  # K5 holds 8 trace blocks per wave: columns of A1 by groups of 128 neurons; C C^T is tiny
                Cl = C[:, :registered.shape[0]] if times is None else C[:, torch.as_tensor(times, device=C.device).long()]
                for s0 in range(0, K, 128):
                    part, _ = ops.spatial_accum(registered, C[s0:s0 + 128].contiguous(), frame_ids=frame_ids, times=times)
                    A1[:, s0:s0 + 128] = part
                Cs.copy_((Cl.double() @ Cl.double().T).float())
        if self.group is not None and torch.distributed.get_world_size(self.group) > 1:
            if self.collective == "c1" and torch.distributed.get_backend(self.group) == "nccl":
                # the library's own RCCL communicator (C1 of the C ABI: what a host without torch would call), built on
                # first use
                if self._comm is None:
                    self._comm = ops.Communicator(self.group)
                self._comm.all_reduce_(buf)
            else:  # the caller's process group: RCCL when its backend is "nccl" (one process per GPU), gloo in rehearsals
                with ops._timed("allreduce"):
                    torch.distributed.all_reduce(buf, group=self.group)
        A2 = fp.A.reshape(P, K).contiguous()
        Dd = None
        if D is not None:   # the fp32 device copy is kept while the caller hands in the same object (105 MB at cfg 3)
            if self._D_dev is None or self._D_dev[0] is not D:
                self._D_dev = (D, torch.as_tensor(D).to(device, torch.float32).reshape(P, K).contiguous())
            Dd = self._D_dev[1]
        if sl is not None:
            ops.mu_spatial_lists(A2, fp.packed_lists(), sl, A1, Cs, fp.sz_list, Dd, gamma)
        else:
            ops.mu_spatial(A2, A1, Cs, Dd, gamma)
        fp.A = A2.view(*fp.sz_list, K)
        fp.invalidate_layouts()   # K6 wrote through the raw pointer: the packed copies are stale
        self._sl = None
        return fp.A

    def _spatial_lists(self):
        """Tile lists of the list-form footprint update (``ops.spatial_lists_setup``) when ``spatial_kernel`` allows it and
        the footprints are compact (the rule of the Gram kernel: few boxes per voxel; no tile with more than 32 neurons),
        else None: dense K5 / K6."""
        fp = self.fp
        if self.spatial_kernel not in ('auto', 'lists') or fp.K > 256 or fp.P * 32 >= 2 ** 31 or fp.sz_list[1] * fp.sz_list[2] < 4:
            return None
        ly = fp.packed_lists()
        if self.spatial_kernel == 'auto' and ly["boxfrac"] >= LISTS_BOXFRAC_LIMIT:
            return None
        key = (fp.A.data_ptr(), fp.A._version)
        if getattr(self, "_sl", None) is None or self._sl[0] != key:
            self._sl = (key, ops.spatial_lists_setup(ly, fp.K, fp.sz_list))
        sl = self._sl[1]
        if sl["total"] <= 0:
            if self.spatial_kernel == 'lists':
                raise ValueError("spatial_kernel='lists': a tile lists more than 32 neurons (or none at all)")
            return None
        return sl

    # ---- fit steps -------------------------------------------------------------------------------------
    def _gather_frames(self, loader):
        """All frames the loader yields, in its order, resident on the GPU as (T,P) plus their indices."""
        if isinstance(loader, ResidentLoader):
            return loader.frames_2d(), loader.order_tensor()
        staged = self._stage_epoch(loader) if self.stream_loader else None
        if staged is not None:
            order = torch.tensor([t for b in staged[1] for t in b], dtype=torch.int32)
            frames = staged[0]
            if staged[2]:   # the dataset's device frames, row t = frame t: in the loader's order
                n = order.numel()
                if n != frames.shape[0] or not bool((order == torch.arange(n, dtype=torch.int32)).all()):
                    frames = frames.index_select(0, order.to(device, torch.int64))
            return frames, order.to(device)
        fr, idx = [], []
        for data in loader:
            fr.append(data[0].to(device, torch.float32).reshape(data[0].shape[0], -1))
            idx.append(torch.as_tensor(data[1]).to(device, torch.int32).reshape(-1))
        return torch.cat(fr, 0).contiguous(), torch.cat(idx, 0)

    def update_footprints(self, testloader, batch_size, sz, gamma_c=1e-2, gamma_a=1e0, iter_c=10, return_dense=None,
                          live_spatial=False, iter_a=1):
        """Reference :163-179: ``iter_c`` multiplicative updates of ``self.C`` under the current warp.

        Returns ``(A_t, Y_i, Y)`` like the reference when the dense float64 ``A_t`` fits
        ``DENSE_RETURN_LIMIT`` (or ``return_dense=True``); otherwise ``(None, None, None)``.
        ``Y_i`` comes from the nearest-neighbour search K7 (``dnmf_image_iwarp``).
        ``gamma_a`` is unused unless ``live_spatial`` is set, as in the reference (its footprint update is commented
        out, :174).  ``live_spatial=True`` (extension) wires that update: after the temporal updates the frames are
        registered (K7) and ``fp.A`` takes ``iter_a`` multiplicative updates ``A * (Y_i C^T) / (A C C^T + gamma_a D +
        1e-32)`` (``spatial_step``: K5, one all-reduce over ``self.group``, K6) with the ``D`` of the constructor
        flattened over the voxels -- the reference's commented lines update an unrelated random ``self.A`` with
        mismatched shapes (:131, :169-176); here the update acts on the footprints the model uses."""
        fp = self.fp
        K, P = fp.K, fp.P
        with torch.no_grad():
            frames, order = self._gather_frames(testloader)
            T_loc = frames.shape[0]
            Csel = self.C.to(device, torch.float32)[:, order.long()].contiguous()
            ly = self._lists_layout_for_fused_update(gamma_c)
            if ly is not None:
                # K3n leaves its slot tables in the workspace and K4 reads them there: no dense (T,K,K) in between
                _, _, self._ws_k3 = ops.warp_gram_rhs_lists(ly, K, fp.sz_list, fp.beta.detach(), order, frames,
                                                            workspace=self._ws_k3, finish=False)
                Cnew = ops.mu_temporal_slots(ly, self._ws_k3, fp.sz_list, Csel.clone(), iter_c)
            else:
                G, r = self._gram_rhs(frames, order)
                Cnew = _mu_temporal(G, r, Csel, gamma_c, iter_c, group=self.group, nbr=self._gram_nbr)
            C = self.C.to(device, torch.float32).clone()
            C[:, order.long()] = Cnew
            self.C = C
            if live_spatial:
                if self._reg_buf is None or self._reg_buf.shape != (T_loc, P):
                    self._reg_buf = None
                    self._reg_buf = torch.empty((T_loc, P), dtype=torch.float32, device=device)
                ops.image_iwarp(frames, None, fp.sz_list, fp.beta.detach(), order, out=self._reg_buf)
                for _ in range(iter_a):
                    self.spatial_step(self._reg_buf, D=self.D, gamma=gamma_a, times=order)
            if return_dense is None:
                return_dense = 8 * P * K * T_loc <= DENSE_RETURN_LIMIT
            if not return_dense:
                return None, None, None
            X, Y_, Z = fp.sz_list
            A_t = np.empty((X, Y_, Z, K, T_loc))
            step = max(1, int((256 << 20) // (4 * P * K)))
            for s in range(0, T_loc, step):
                a, _ = ops.warp_gather(fp.A.contiguous(), fp.beta.detach(), order[s:s + step], want_grid=False)
                A_t[..., s:s + step] = a.permute(2, 3, 4, 1, 0).double().cpu().numpy()
            Yv = frames.view(T_loc, X, Y_, Z).permute(1, 2, 3, 0).double().cpu().numpy()
            Yi = np.empty_like(Yv)
            for s in range(0, T_loc, 256):
                yi = ops.image_iwarp(frames, order[s:s + 256], fp.sz_list, fp.beta.detach(), order[s:s + 256])
                Yi[..., s:s + 256] = yi.view(-1, X, Y_, Z).permute(1, 2, 3, 0).double().cpu().numpy()
            return A_t, Yi, Yv

    def _lists_layout_for_fused_update(self, gamma_c):
        """The K3n layout when update_footprints can run as K3n + K4-on-slots (one channel, no neighbour term, the
        kernel choice allows K3n and the pattern is narrow enough for the per-row lists); else None."""
        fp = self.fp
        if not (gamma_c is None or gamma_c == 0) or len(self._channels()) != 1 or fp.K > 256:
            return None
        if self.gram_kernel not in ('auto', 'lists'):
            return None
        ly = fp.packed_lists()
        if ly["nbr"] is None or ly["nslot"] > ops.LISTS_MAX_SLOTS:
            return None
        return ly if (self.gram_kernel == 'lists' or ly["boxfrac"] < LISTS_BOXFRAC_LIMIT) else None

    def _gram_rhs(self, frames, order):
        """Per-frame Gram matrices and right-hand sides under the current warp, summed over the channels."""
        G = r = None
        chans = self._channels()
        self._gram_nbr = None
        for fp, cols in chans:
            Gc, rc = self._gram_rhs_one(fp, frames if cols is None else frames[:, cols], order)
            G, r = (Gc, rc) if G is None else (G.add_(Gc), r.add_(rc))
        if len(chans) > 1:
            self._gram_nbr = None   # the channels' patterns may differ
        return G, r

    def _channels(self):
        """``[(spatial model, slice of a frame row)]``: one entry, the whole row, for the reference's single-channel
        model; MultiChannelDNMF lists its colour channels here."""
        return [(self.fp, None)]

    def _note_once(self, key, text):
        """One line on stdout the first time ``key`` comes up (kernel choices that cost a factor and would otherwise
        go unnoticed)."""
        if key not in self._warned:
            self._warned.add(key)
            print("[dnmf_amd] " + text, flush=True)

    def _gram_rhs_one(self, fp, frames, order):
        """K3, K3s or K3n on the footprints of ``fp``."""
        if self.gram_kernel == 'auto':
            if fp.K > 256:
                self._note_once("k3n-K", f"gram_kernel='auto': K={fp.K} > 256, the neuron-list kernel K3n does not "
                                "apply; using K3s / K3 by pairs of neuron groups (several times slower)")
            else:
                ly = fp.packed_lists()
                if ly["nslot"] > ops.LISTS_MAX_SLOTS or ly["boxfrac"] >= LISTS_BOXFRAC_LIMIT:
                    self._note_once("k3n-shape", f"gram_kernel='auto': footprints too wide for the neuron-list kernel K3n "
                                    f"(pattern slots {ly['nslot']} vs limit {ops.LISTS_MAX_SLOTS}, mean boxes per voxel "
                                    f"{ly['boxfrac']:.2f} vs limit {LISTS_BOXFRAC_LIMIT}); using K3s / K3 (6-40x slower)")
        if self.gram_kernel in ('auto', 'lists') and fp.K <= 256:
            ly = fp.packed_lists()
            # K3n pays per (voxel, listed neuron) and per listed pair: it wins while a voxel lies in few boxes
            if ly["nslot"] <= ops.LISTS_MAX_SLOTS and (self.gram_kernel == 'lists' or ly["boxfrac"] < LISTS_BOXFRAC_LIMIT):
                G, r, self._ws_k3 = ops.warp_gram_rhs_lists(ly, fp.K, fp.sz_list, fp.beta.detach(), order, frames,
                                                            workspace=self._ws_k3)
                self._gram_nbr = ly["nbr"]   # the pattern of this G, for K4
                return G, r
            if self.gram_kernel == 'lists':
                raise ValueError(f"gram_kernel='lists': the pattern of G has {ly['nslot']} slots > "
                                 f"{ops.LISTS_MAX_SLOTS} (footprints overlap too much)")
        if fp.K > 127:
            return self._gram_rhs_grouped(fp, frames, order)
        sp = fp.packed_sparse() if self.gram_kernel in ('auto', 'sparse') else None
        if sp is not None and (self.gram_kernel == 'sparse' or sp["occupancy"] < 0.5):
            G, r, self._ws_k3 = ops.warp_gram_rhs_sparse(sp["Aps"], fp.K, sp["order"], sp["row_mask"], fp.sz_list,
                                                         fp.beta.detach(), order, frames, workspace=self._ws_k3)
        else:
            G, r, self._ws_k3 = ops.warp_gram_rhs(fp.packed_footprints(), fp.K, fp.sz_list, fp.beta.detach(), order,
                                                  frames, workspace=self._ws_k3, bf16=self.gram_kernel == 'bf16')
        return G, r

    def _gram_rhs_grouped(self, fp, frames, order):
        """K > 127: one launch holds at most 8 blocks of 16 channels, so the neurons are cut into groups and every
        PAIR of groups is one launch on their union; the launch yields both diagonal blocks and the off-diagonal
        block of the pair (diagonal blocks are recomputed by every pair they belong to).  K3s (groups of 64 along
        the Z-order curve) unless ``gram_kernel == 'dense'`` or the footprints are dense (K3, groups of 56)."""
        K, B = fp.K, order.numel()
        G = torch.empty((B, K, K), dtype=torch.float32, device=device)
        r = torch.empty((B, K), dtype=torch.float32, device=device)
        if self.gram_kernel in ('auto', 'sparse'):
            pairs = fp.packed_sparse_pairs()
            if self.gram_kernel == 'sparse' or max(sp["occupancy"] for _, sp in pairs) < 0.5:
                for cols, sp in pairs:
                    Gp, rp, self._ws_k3 = ops.warp_gram_rhs_sparse(sp["Aps"], cols.numel(), sp["order"], sp["row_mask"],
                                                                   fp.sz_list, fp.beta.detach(), order, frames,
                                                                   workspace=self._ws_k3)
                    G[:, cols[:, None], cols[None, :]] = Gp
                    r[:, cols] = rp
                return G, r
        group = 56
        groups = [list(range(s0, min(K, s0 + group))) for s0 in range(0, K, group)]
        for i in range(len(groups)):
            for j in range(i + 1, len(groups)):
                cols = groups[i] + groups[j]
                Gp, rp, self._ws_k3 = ops.warp_gram_rhs(fp.packed_columns(cols), len(cols), fp.sz_list, fp.beta.detach(),
                                                        order, frames, workspace=self._ws_k3,
                                                        bf16=self.gram_kernel == 'bf16')
                idx = torch.as_tensor(cols, device=device)
                G[:, idx[:, None], idx[None, :]] = Gp
                r[:, idx] = rp
        return G, r

    def _recon_cache(self):
        """Reconstruction images S_t = A.C_t of all T frames, one (T,lds) tensor per channel (C is constant inside
        update_motion); None when they do not fit ``RECON_CACHE_LIMIT``."""
        chans = self._channels()
        fp0 = chans[0][0]
        lds = ops.halo_voxels(fp0.sz_list)
        if 4 * lds * fp0.T * len(chans) > RECON_CACHE_LIMIT:
            return None
        C = self.C.to(device, torch.float32).contiguous()
        all_t = torch.arange(fp0.T, dtype=torch.int32, device=device)
        # the buffers are kept between calls: re-allocating gigabytes every epoch makes the caching allocator split and
        # re-map its large blocks (tens of milliseconds every few sweeps)
        if self._S_bufs is None or len(self._S_bufs) != len(chans) or self._S_bufs[0].shape != (fp0.T, lds):
            self._S_bufs = [torch.empty((fp0.T, lds), dtype=torch.float32, device=device) for _ in chans]
            self._S_zero = [[None] for _ in chans]
        out = []
        for (fp, _), S, zs in zip(chans, self._S_bufs, self._S_zero):
            # every row is rewritten on every call; tiles no neuron reaches were zeroed by the first call with these
            # footprints and are skipped from then on (a third of the tiles -- and of the stores -- at 512x512, K = 100)
            was = zs[0]
            for s in range(0, fp.T, 32768):
                state = [was]
                fp.recon_image(C, all_t[s:s + 32768], out=S[s:s + 32768], zero_state=state)
            zs[0] = state[0]
            out.append(S)
        return out

    def _k2(self, S_all, Cdev, frames, frame_ids, times, grad, norm, want):
        """K2 for one set of frames on every channel: ``grad`` (10,3,T) is incremented by d mse / d beta, the mean
        running over ``norm`` frames (0 = all of ``times``) x channels x voxels.  Returns the outputs of
        ``ops.warp_recon_grad`` with the losses summed over channels."""
        chans = self._channels()
        nc = len(chans)
        if (nc > 1 or times.numel() > K2_MAX_FRAMES) and norm == 0:
            norm = times.numel()
        if times.numel() > K2_MAX_FRAMES:
            # frames ride on gridDim.y (<= 65535): larger sets go in pieces; grad accumulates, columns are independent
            total = None
            for s in range(0, times.numel(), K2_MAX_FRAMES):
                e = s + K2_MAX_FRAMES
                out = self._k2(S_all, Cdev, frames if frame_ids is not None else frames[s:e],
                               None if frame_ids is None else frame_ids[s:e], times[s:e], grad, norm, want)
                if total is None:
                    total = out
                elif want:
                    total["loss"] = total["loss"] + out["loss"]
                    total["frame_loss"] = torch.cat((total["frame_loss"], out["frame_loss"]))
                    total["reg"] = torch.cat((total["reg"], out["reg"]))
            return total
        total = None
        for c, (fp, cols) in enumerate(chans):
            if S_all is not None:
                S, s_ids = S_all[c], times
            else:
                S, s_ids = fp.recon_image(Cdev, times), None
            out = ops.warp_recon_grad(S, s_ids, frames if cols is None else frames[:, cols], frame_ids, fp.sz_list,
                                      fp.beta.detach(), times, grad=grad, want_loss=want, want_reg=want,
                                      workspace=self._ws_k2, norm_frames=norm * nc)
            self._ws_k2 = out["workspace"]
            if total is None:
                total = out
            elif want:
                total["loss"] += out["loss"]
                total["frame_loss"] += out["frame_loss"]
        return total

    def update_motion(self, dataloader, optimizer, gamma=0, epochs=20):
        """Reference :181-194: mini-batch steps of the caller's optimiser on ``fp.beta`` against
        ``mse(A_tC, frames) + gamma*mean(reg)`` (the reg term is gradient-free in the reference, :60-61).

        A ``ResidentLoader`` hands over rows that already live on the GPU.  Any other loader (the stock
        ``torch.utils.data.DataLoader`` of demo.py:33-35) is iterated once per epoch by a background thread while this
        thread copies the mini-batches into a device buffer (``_stage_epoch``); the epoch then runs from that buffer --
        as four launches when the optimiser is the demo's plain Adam (``_motion_epoch``), else step by step."""
        fp = self.fp
        beta = fp.beta
        # reconstruction images of all frames (C is constant inside this call): built when first needed -- the fused
        # epoch on compact footprints makes its own, chunk by chunk (_motion_epoch)
        cache = {}

        def recon():
            if not cache:
                cache["S"] = self._recon_cache()
                cache["C"] = None if cache["S"] is not None else self.C.to(device, torch.float32).contiguous()
            return cache["S"], cache["C"]

        chunked = self.motion_chunk > 0 and self._motion_lists_layout() is not None
        resident = isinstance(dataloader, ResidentLoader)
        for epoch in range(1, epochs + 1):
            if self.verbose:
                print('Epoch ' + str(epoch))
            fp.train()
            if resident:
                plan, frames, rows = dataloader.epoch_plan(), dataloader.frames_2d(), None
                norms = [min(dataloader.batch_size, dataloader.T_total - j * dataloader.batch_size)
                         for j in range(plan.nsteps)]
            else:
                staged = self._stage_epoch(dataloader) if self.stream_loader else None
                if staged is None:      # unknown length or too large to stage: batch by batch from the host
                    self._motion_epoch_from_host(dataloader, optimizer, *recon())
                    continue
                frames, batch_times, by_frame = staged
                plan = sharding.plan_from_batches(batch_times, fp.T)
                norms = [len(b) for b in batch_times]
                rows = None     # frame t in row t (a dataset's own device frames)
                if not by_frame:
                    # row of the staging buffer that holds frame t
                    flat = torch.tensor([t for b in batch_times for t in b], dtype=torch.int64)
                    rows = torch.zeros(fp.T, dtype=torch.int32)
                    rows[flat] = torch.arange(flat.numel(), dtype=torch.int32)
                    rows = rows.to(device)
                if plan is None:        # a frame served twice in one epoch: the steps as they come
                    plan = sharding.EpochPlan(len(batch_times), None, None, None, [torch.tensor(b, dtype=torch.int64)
                                                                                   for b in batch_times])
            if plan.frame_step is not None and self._fusable(optimizer) and (chunked or recon()[0] is not None):
                self._motion_epoch(plan, frames, rows, optimizer, None if chunked else recon()[0])
                continue
            S_all, Cdev = recon()
            for batch_idx, b in enumerate(plan.batches):
                optimizer.zero_grad()
                if beta.grad is None:
                    beta.grad = torch.zeros_like(beta)
                if b.numel() == 0:      # sharded: this global mini-batch has no frame here; still a step
                    optimizer.step()
                    continue
                times = b.to(device, torch.int32)
                want = self.verbose and batch_idx % 10 == 0
                out = self._k2(S_all, Cdev, frames, times if rows is None else rows[times.long()], times, beta.grad,
                               norms[batch_idx], want)
                optimizer.step()
                if want:
                    print('Recon: ' + str(out["loss"][0]))
                    print('Reg: ' + str(out["reg"]))

    def _motion_epoch_from_host(self, dataloader, optimizer, S_all, Cdev):
        """One epoch with every mini-batch copied from the host when it is needed (loaders that cannot be staged)."""
        beta = self.fp.beta
        for batch_idx, data in enumerate(dataloader):
            times = torch.as_tensor(data[1]).to(device, torch.int32).reshape(-1)
            frames = data[0].to(device, torch.float32).reshape(times.numel(), -1)
            optimizer.zero_grad()
            if beta.grad is None:
                beta.grad = torch.zeros_like(beta)
            want = self.verbose and batch_idx % 10 == 0
            out = self._k2(S_all, Cdev, frames, None, times, beta.grad, 0, want)
            optimizer.step()
            if want:
                print('Recon: ' + str(out["loss"][0]))
                print('Reg: ' + str(out["reg"]))

    @staticmethod
    def _loader_frames(loader):
        """Frames one pass over ``loader`` yields, or None when that cannot be told without iterating it."""
        sampler = getattr(loader, "sampler", None)
        if getattr(loader, "dataset", None) is not None and sampler is not None and hasattr(sampler, "__len__") \
                and getattr(loader, "batch_size", None):
            n = len(sampler)
            return n // loader.batch_size * loader.batch_size if getattr(loader, "drop_last", False) else n
        if isinstance(loader, (list, tuple)):
            return sum(len(d[1]) for d in loader)
        return None

    def _stage_epoch(self, loader):
        """One pass over a host loader into a device buffer: returns ``(frames (n, row) fp32 CUDA rows in arrival
        order, [frame indices of every mini-batch])``, or None when the number of frames is unknown or the buffer would
        exceed ``STAGE_LIMIT``.  The loader is iterated by a background thread (its collate copies overlap the
        host-to-device copies issued here; pinned batches -- ``DataLoader(pin_memory=True)`` -- are copied
        asynchronously), and the buffer is kept between calls."""
        n = self._loader_frames(loader)
        row = sum(f.P for f, _ in self._channels())
        res = self._resident_batches(loader, row)
        if res is not None:
            return res[0], res[1], True
        if n is None or n == 0 or 4 * n * row > STAGE_LIMIT:
            return None
        if self._stage_buf is None or self._stage_buf.shape[0] < n or self._stage_buf.shape[1] != row:
            self._stage_buf = None
            self._stage_buf = torch.empty((n, row), dtype=torch.float32, device=device)
        buf = self._stage_buf
        batch_times, at = [], 0
        for data in _Prefetch(loader):
            idx = data[1].tolist() if hasattr(data[1], "tolist") else [int(i) for i in data[1]]
            if isinstance(idx, int):
                idx = [idx]
            B = len(idx)
            if at + B > n:
                raise RuntimeError(f"the loader yielded more than the {n} frames its length announces")
            src = data[0].reshape(B, -1)
            if src.shape[1] != row:
                raise ValueError(f"a frame has {src.shape[1]} values, the model expects {row}")
            buf[at:at + B].copy_(src, non_blocking=src.is_pinned() if hasattr(src, "is_pinned") else False)
            batch_times.append(idx)
            at += B
        return buf[:at], batch_times, False

    @staticmethod
    def _resident_batches(loader, row):
        """``(frames, index batches)`` of one pass over a stock single-process ``DataLoader`` whose dataset keeps its
        frames on the GPU (``dataset.device_frames()``: row t = frame t), without fetching a single sample: the
        loader's iterator is created as for an ordinary pass -- it draws the seeds an ordinary pass draws, so shuffled
        orders are the ones the reference loop would see -- and only its index batches are pulled.  None when the
        loader is anything else."""
        from torch.utils.data import DataLoader
        from torch.utils.data._utils.collate import default_collate
        from torch.utils.data.dataloader import _BaseDataLoaderIter
        if type(loader) is not DataLoader or loader.num_workers != 0 or loader.batch_sampler is None \
                or loader.collate_fn is not default_collate or not hasattr(_BaseDataLoaderIter, "_next_index"):
            return None
        get = getattr(loader.dataset, "device_frames", None)
        if get is None:
            return None
        frames = get()
        if frames.dim() != 2 or frames.shape[1] != row:
            return None
        it = iter(loader)
        batches = []
        while True:
            try:
                batches.append([int(i) for i in it._next_index()])
            except StopIteration:
                break
        return frames, batches

    def _fusable(self, optimizer):
        """True when ``optimizer`` is exactly the reference demo's: torch.optim.Adam([fp.beta]) without
        amsgrad / weight decay / maximize, so that its update can be evaluated per column in closed form."""
        if not self.fused_motion or type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            return False
        g = optimizer.param_groups[0]
        if len(g['params']) != 1 or g['params'][0] is not self.fp.beta:
            return False
        return not (g.get('amsgrad') or g.get('maximize') or g.get('weight_decay', 0) != 0 or g.get('capturable')
                    or g.get('differentiable') or g.get('fused') or g.get('decoupled_weight_decay'))

    def _motion_lists_layout(self):
        """The K3n layout when the motion gradient can take its reconstruction images from the neuron lists (one
        channel, compact footprints); else None."""
        fp = self.fp
        if len(self._channels()) != 1 or not fp.use_lists or fp.K > 256:
            return None
        ly = fp.packed_lists()
        return ly if ly["boxfrac"] < LISTS_BOXFRAC_LIMIT else None

    def _motion_epoch(self, plan, frames, rows, optimizer, S_all):
        """One epoch of mini-batch Adam steps: dnmf_adam_epoch phase 0, the gradient of every frame at its coasted
        beta, phase 1.  ``plan``: the epoch's EpochPlan; ``frames`` (n, row) device rows, frame t in row ``rows[t]``
        (``rows`` None: in row t).  ``S_all`` None: compact footprints, the reconstruction images are made
        ``motion_chunk`` frames at a time right before K2 gathers from them (``dnmf_motion_grad_lists``: they stay in
        the Infinity Cache); else the cached images of all frames and one K2 launch per group of equal mini-batch
        size.  The optimiser's own state tensors are read and written, so the caller's optimiser stays valid and a
        later un-fused step continues from it."""
        fp, beta = self.fp, self.fp.beta
        g = optimizer.param_groups[0]
        state = optimizer.state[beta]
        if len(state) == 0:  # what torch.optim.Adam._init_group creates on the first step()
            state['step'] = torch.tensor(0.0, dtype=torch.get_default_dtype())
            state['exp_avg'] = torch.zeros_like(beta, memory_format=torch.preserve_format)
            state['exp_avg_sq'] = torch.zeros_like(beta, memory_format=torch.preserve_format)
        n = plan.nsteps
        step0 = int(state['step'])
        args = (step0, plan.frame_step.to(device), n, g['lr'], g['betas'], g['eps'])
        order = plan.order.to(device)
        with torch.no_grad():
            ops.adam_epoch(beta, None, state['exp_avg'], state['exp_avg_sq'], *args, phase=0, order=order)
            grad = torch.zeros_like(beta)
            outs = []
            Cdev = self.C.to(device, torch.float32).contiguous() if S_all is None else None
            for idx, nf in plan.groups:
                if idx.numel() == 0:
                    continue
                idx = idx.to(device, torch.int32)
                fid = idx if rows is None else rows[idx.long()]
                if S_all is None:
                    out = ops.motion_grad_lists(self._motion_lists_layout(), fp.K, fp.sz_list, Cdev, frames, fid,
                                                beta.detach(), idx, grad, nf, self.motion_chunk, want=self.verbose,
                                                workspace=self._ws_mg)
                    self._ws_mg = out["workspace"]
                else:
                    out = self._k2(S_all, None, frames, fid, idx, grad, nf, self.verbose)
                outs.append((idx, nf, out))
            ops.adam_epoch(beta, grad, state['exp_avg'], state['exp_avg_sq'], *args, phase=1, order=order)
        state['step'] += n
        beta.grad = grad
        if self.verbose and outs:
            idx, nf, out = outs[0]
            for j in range(0, idx.numel() // nf, 10):
                print('Recon: ' + str(out["frame_loss"][j * nf:(j + 1) * nf].sum()))
                print('Reg: ' + str(out["reg"][j * nf:(j + 1) * nf]))

    def fit(self, dataloader, testloader, optimizer, batch_size, outer=5, gamma=1, epochs=10, gamma_c=0, iter_c=50,
            spatial=False, gamma_a=1e0):
        """Convenience wrapper of the loop ``demo.py:44-46`` writes out (not part of the reference).  ``spatial=True``
        also updates the footprints after every temporal update (``update_footprints(live_spatial=True)``)."""
        out = (None, None, None)
        for _ in range(outer):
            self.update_motion(dataloader, optimizer, gamma=gamma, epochs=epochs)
            out = self.update_footprints(testloader, batch_size, self.fp.sz_list, gamma_c=gamma_c, gamma_a=gamma_a,
                                         iter_c=iter_c, live_spatial=spatial)
        return out


class MultiChannelDNMF(DeformableNMF):
    """Several colour channels of one volume (BASELINE config 5, "C=3").  NOT in the reference, which has no channel
    axis on this path: the channels are treated as extra voxels that share the warp ``beta`` and the
    traces ``C`` -- channel ``c`` shows neuron ``k`` with footprint ``colours[c,k] * A[...,k]`` -- so every update
    formula of the reference holds with the sums over voxels also running over channels:
    ``G_t = sum_c A_t(c)^T A_t(c)``, ``r_t = sum_c A_t(c)^T y_t(c)``, loss = mean over batch x channels x voxels.

    A frame is the concatenation of its channels, ``(C, X, Y, Z)``: loaders yield ``(B, C, X, Y, Z)`` and a
    ``ResidentLoader`` is built on rows of ``C*P`` floats.  ``fp`` holds the uncoloured footprints and ``beta``."""

    def __init__(self, sz, K, T, colours, positions=None):
        super().__init__(sz, K, T, positions)
        colours = torch.as_tensor(colours, dtype=torch.float32).to(device)
        if colours.dim() != 2 or colours.shape[1] != K:
            raise ValueError(f"colours must be (channels, K={K}), got {tuple(colours.shape)}")
        self.colours = colours
        self._chan_fp = None
        self._chan_key = None

    def _channels(self):
        import copy
        key = (self.fp.A.data_ptr(), self.fp.A._version, self.colours.data_ptr(), self.colours._version)
        if self._chan_fp is None or self._chan_key != key:
            P = self.fp.P
            self._chan_fp = []
            for c in range(self.colours.shape[0]):
                f = copy.copy(self.fp)  # shares beta (the caller's optimiser steps one tensor) and the lattice
                f.A = self.fp.A * self.colours[c]
                f._packed = f._sparse = f._sparse_pairs = f._lists = None
                self._chan_fp.append((f, slice(c * P, (c + 1) * P)))
            self._chan_key = key
        return self._chan_fp

    def update_footprints(self, testloader, batch_size, sz, gamma_c=1e-2, gamma_a=1e0, iter_c=10, return_dense=False):
        """As DeformableNMF.update_footprints on the channel sums; the dense ``A_t`` return is not offered."""
        if return_dense:
            raise NotImplementedError("MultiChannelDNMF.update_footprints: return_dense")
        return super().update_footprints(testloader, batch_size, sz, gamma_c=gamma_c, gamma_a=gamma_a, iter_c=iter_c,
                                         return_dense=False)

    def spatial_step(self, registered, D=None, gamma=None, frame_ids=None, times=None):
        """One multiplicative update of the (uncoloured) footprints ``fp.A`` from the registered frames of ALL channels
        (rows of NC * P floats).  With channel c showing neuron k as ``colours[c,k] A[:,k]`` the update of the shared ``A``
        that the reference's formula (``Demix/dNMF.py:151-160``) becomes is

            A <- A * (sum_c colours_c (Y_i^c C^T)) / (A (C C^T * colours^T colours) + gamma D + 1e-32)

        (numerator: K5 per channel, scaled by the channel's colours; denominator: K6 with C_s multiplied entry by entry by
        the Gram matrix of the colours).  One all-reduce of the A1 | C_s buffer over ``self.group`` like the single-channel
        step.  Not in the reference (no channel axis there): checked against this formula in float64 and, for one channel of
        colour 1, against ``DeformableNMF.spatial_step``."""
        fp = self.fp
        P, K = fp.P, fp.K
        NC = self.colours.shape[0]
        if registered.shape[1] != NC * P:
            raise ValueError(f"MultiChannelDNMF.spatial_step: rows of {registered.shape[1]} floats, expected {NC} x {P}")
        if K > 128:
            raise NotImplementedError("MultiChannelDNMF.spatial_step: K > 128")
        C = self.C.to(device, torch.float32).contiguous()
        if times is None:
            times = frame_ids
        if self._spatial_buf is None or self._spatial_buf.numel() != P * K + K * K:
            self._spatial_buf = None
            self._spatial_buf = torch.empty((P * K + K * K,), dtype=torch.float32, device=device)
        buf = self._spatial_buf
        A1, Cs = buf[:P * K].view(P, K), buf[P * K:].view(K, K)
        A1.zero_()
        for c in range(NC):
            part, cs = ops.spatial_accum(registered[:, c * P:(c + 1) * P], C, frame_ids=frame_ids, times=times)
            A1.addcmul_(part, self.colours[c][None, :])
            if c == 0:
                Cs.copy_(cs)
        if self.group is not None and torch.distributed.get_world_size(self.group) > 1:
            with ops._timed("allreduce"):
                torch.distributed.all_reduce(buf, group=self.group)
        Cs.mul_(self.colours.T @ self.colours)
        A2 = fp.A.reshape(P, K).contiguous()
        Dd = None if D is None else torch.as_tensor(D).to(device, torch.float32).reshape(P, K).contiguous()
        ops.mu_spatial(A2, A1, Cs, Dd, gamma)
        fp.A = A2.view(*fp.sz_list, K)
        fp.invalidate_layouts()
        self._chan_fp = None
        return fp.A


def _mu_temporal(G, r, C, gamma, iters, group=None, nbr=None):
    """``iters`` multiplicative updates on (T,K,K) / (T,K) Gram data.

    ``C`` fp32 (K,T): the state update_footprints starts from (the reference's ``self.C``); without the
    neighbour term the whole loop is one K4 launch (fp64 inside, one rounding to fp32 at the end, as
    reference :177).  ``C`` fp64, or gamma != 0: the fp64 state is iterated with the one-round kernel.
    ``group``: torch.distributed group when the frames are a contiguous T-shard of rank order: with gamma != 0
    every round first exchanges the boundary columns (one all-gather of 2K doubles) so that the first / last frame
    of a shard sees its true neighbour instead of the replicated edge.  Returns a tensor of C's dtype."""
    if C.dtype == torch.float32 and (gamma is None or gamma == 0):
        return ops.mu_temporal(G, r, C.contiguous().clone(), iters, nbr=nbr)
    a = C.double().contiguous().clone()
    b = torch.empty_like(a)
    sharded = group is not None and gamma is not None and gamma != 0 and torch.distributed.get_world_size(group) > 1
    if sharded:
        rank, world = torch.distributed.get_rank(group), torch.distributed.get_world_size(group)
        edges = [torch.empty((2, a.shape[0]), dtype=torch.float64, device=a.device) for _ in range(world)]
    for _ in range(iters):
        left = right = None
        if sharded:
            mine = torch.stack((a[:, 0], a[:, -1])).contiguous()
            torch.distributed.all_gather(edges, mine, group=group)
            left = edges[rank - 1][1].contiguous() if rank > 0 else None
            right = edges[rank + 1][0].contiguous() if rank + 1 < world else None
        ops.mu_temporal_step(G, r, a, b, 0.0 if gamma is None else float(gamma), left, right)
        a, b = b, a
    return a.to(C.dtype)


class _Prefetch:
    """Iterates ``loader`` in a background thread, two items ahead: the loader's own work (indexing the dataset,
    collating a mini-batch: host memory copies that release the GIL) overlaps what the consumer does with the
    previous item.  Exceptions of the loader are re-raised in the consumer."""

    _END = object()

    def __init__(self, loader, depth=2):
        import queue
        import threading
        self._q = queue.Queue(maxsize=depth)
        self._stop = False
        self._thread = threading.Thread(target=self._run, args=(loader,), daemon=True)
        self._thread.start()

    def _put(self, item):
        import queue
        while not self._stop:
            try:
                self._q.put(item, timeout=0.1)
                return
            except queue.Full:
                pass

    def _run(self, loader):
        try:
            for item in loader:
                self._put(item)
                if self._stop:
                    return
            self._put(self._END)
        except BaseException as exc:  # noqa: BLE001 - handed to the consumer
            self._put(exc)

    def __iter__(self):
        try:
            while True:
                item = self._q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            self._stop = True


class ResidentLoader:
    """Iterates mini-batches of a video that already lives on the GPU as (T,P) rows.

    Yields ``(frames (B,X,Y,Z), idx int32 tensor)`` like a DataLoader over SimulatedVideoDataset; the fit steps
    recognise it and index the resident rows directly instead of copying batches (the timed region of bench.py
    starts with the inputs in HBM).  ``shuffle=True`` draws a new permutation per epoch from ``generator``.

    Sharded use (one process per GPU): ``frames`` holds the block ``[t0, t0+T_local)`` of a ``T_total``-frame
    video.  Every rank passes a generator with the SAME seed; mini-batches are cut from the global permutation
    and each rank keeps its own members (``dnmf_amd/sharding.py``), so the optimiser-step sequence is the
    single-process one."""

    def __init__(self, frames, sz, batch_size, shuffle=False, generator=None, t0=0, T_total=None):
        self.sz = _sz_list(sz)
        self._frames = frames.to(device, torch.float32).reshape(frames.shape[0], -1).contiguous()
        self.batch_size, self.shuffle, self.generator = int(batch_size), shuffle, generator
        self.T = self._frames.shape[0]
        self.t0 = int(t0)
        self.T_total = self.T if T_total is None else int(T_total)

    def frames_2d(self):
        return self._frames

    def order_tensor(self):
        return torch.arange(self.T, dtype=torch.int32, device=device)

    def __len__(self):
        return (self.T_total + self.batch_size - 1) // self.batch_size

    def epoch_plan(self):
        """Draw this epoch's (global) frame order and return its EpochPlan for the local block."""
        perm = torch.randperm(self.T_total, generator=self.generator) if self.shuffle else torch.arange(self.T_total)
        return sharding.plan_epoch(perm, self.batch_size, self.t0, self.t0 + self.T)

    def iter_indices(self):
        """Local frame indices of each (global) mini-batch, int32 on the GPU; empty for a mini-batch that has no
        frame in this block."""
        for b in self.epoch_plan().batches:
            yield b.to(device, torch.int32)

    def __iter__(self):
        for idx in self.iter_indices():
            yield self._frames[idx.long()].view(-1, *self.sz), idx


class SimulatedVideoDataset(Dataset):
    """Reference ``Demix/dNMF.py:196-217``: a synthetic video and its ground truth."""

    def __init__(self, K, T, sz, shape_std, density, bg_snr, traces, motion, motion_par, resident=False):
        """``resident=True`` (extension) renders the video on the GPU (``dnmf_render_frames``) and keeps it there,
        frame-major; ``video`` is then a CUDA view of shape (X,Y,Z,T) and ``loader()`` hands the rows to the fit
        steps without copies.  The noise is then drawn by the device generator, not the CPU one."""
        if resident:
            if traces != 'exp' or motion != 'gp':
                raise NotImplementedError("only traces='exp', motion='gp' are available")
            frames, positions, traces = Simulator.generate_video_resident(K, T, _sz_list(sz), shape_std, density, bg_snr,
                                                                          motion_par, device=device)
            video = frames.view(T, *_sz_list(sz)).permute(1, 2, 3, 0)
        else:
            video, positions, traces = Simulator.generate_video(K, T, _sz_list(sz), shape_std, density, bg_snr, traces,
                                                                motion, motion_par)
            # same values and the reference's (X,Y,Z,T) shape, but frame-major in memory: ``video[..., t]`` is then
            # one contiguous block instead of P floats that lie T apart (a cache line each)
            video = video.float().permute(3, 0, 1, 2).contiguous().permute(1, 2, 3, 0)
        self.video = video.float()
        self.positions = positions
        self.traces = traces
        self._sz = _sz_list(sz)

    def loader(self, batch_size, shuffle=False, generator=None):
        """A ResidentLoader over this video (clamped at 0 like ``__getitem__`` does frame by frame)."""
        frames = self.video.permute(3, 0, 1, 2).reshape(self.video.shape[3], -1)
        frames = frames.to(device).clamp_(min=0)
        return ResidentLoader(frames, self._sz, batch_size, shuffle=shuffle, generator=generator)

    def device_frames(self):
        """(T, P) fp32 rows on the GPU, row t = what ``self[t][0]`` returns (clamped at 0).  The fit steps take the
        frames of a stock ``DataLoader`` over this dataset from here instead of fetching and collating them one by one
        on the host (~80 us per 512x512 frame: 150 times the GPU work of a sweep); the stored video is clamped in place
        as a pass of ``__getitem__`` calls over all frames would leave it.  The copy is renewed when the stored
        video has been written to since."""
        v = self.video
        cached = getattr(self, "_device_frames", None)
        if cached is not None and cached[0] is v and cached[1] == v._version:
            return cached[2]
        v.clamp_(min=0)
        frames = v.permute(3, 0, 1, 2).reshape(v.shape[3], -1).to(device, torch.float32).contiguous()
        self._device_frames = (v, v._version, frames)
        return frames

    def __len__(self):
        return self.video.shape[3]

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        sample = self.video[:, :, :, idx]
        sample[sample < 0] = 0  # in place on the stored video, like the reference (:214-215)
        return sample, idx


class NeuroPALVideoDataset(Dataset):
    """Reference ``Demix/dNMF.py:220-248``: a recorded video (``data.mat``: ``data`` (X,Y,Z,T)) with tracked
    neuron positions (``traces_n.mat``: ``positions`` (K,3,T), 1-based; ``neuron_names``).  Same fixed
    sub-sampling as the reference (every 2nd voxel in x and y, every 10th in z, the first 100 frames); paths are
    joined portably (the reference hard-codes Windows separators).  The bodies of ``__init__`` and ``__getitem__`` restate
    reference ``Demix/dNMF.py:221-248`` almost line for line (``os.path.join`` instead of ``'\\'``): the file format, the
    sub-sampling and the rescaling of the positions are the interface, there is nothing to redesign in them."""

    def __init__(self, file):
        import os
        from scipy.io import loadmat
        vid_mat = loadmat(os.path.join(file, 'data.mat'))
        self.video = np.array(vid_mat['data'][::2, ::2, ::10, :100]).astype(np.float32)
        pos_mat = loadmat(os.path.join(file, 'traces_n.mat'))
        self.positions = torch.tensor(pos_mat['positions']).float() - 1
        self.positions[:, 0, :] /= 2
        self.positions[:, 1, :] /= 2
        self.positions[:, 2, :] /= 10
        self.names = pos_mat['neuron_names'][0]

    def device_frames(self):
        """(T, P) fp32 rows on the GPU, row t = what ``self[t][0]`` returns; see ``SimulatedVideoDataset.device_frames``.
        The video is a numpy array without a write counter, so the copy is made anew at every call (a pass over the
        100 frames the reference keeps)."""
        np.maximum(self.video, 0, out=self.video)
        T = self.video.shape[3]
        return torch.from_numpy(np.ascontiguousarray(np.moveaxis(self.video, 3, 0)).reshape(T, -1)).to(device, torch.float32)

    def __len__(self):
        return self.video.shape[3]

    def __getitem__(self, idx):
        if torch.is_tensor(idx):
            idx = idx.tolist()
        sample = self.video[:, :, :, idx]
        sample[sample < 0] = 0
        return sample, idx
