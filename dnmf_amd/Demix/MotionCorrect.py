"""Position initialiser: the part of the reference's ``Demix/MotionCorrect.py`` (a vendored copy of CaImAn / NoRMCorre
that nothing in the reference tree calls) that turns a 3-D video into per-frame neuron positions:

    mc = MotionCorrect(video, max_shifts=(6, 6, 1), strides=(24, 24, 1), overlaps=(8, 8, 1), is3D=True)
    mc.motion_correct_pwrigid(template=template)         # fills x_shifts_els, y_shifts_els, z_shifts_els
    positions = mc.apply_shifts_points(video, points)    # (K, 3, T): what DeformableNMF / a tracker starts from

Same constructor arguments, attribute names and method names as the reference class (``MotionCorrect.py:64-385``), limited
to what ``apply_shifts_points`` (``:351-371``) needs: the per-patch shifts of the 3-D piecewise-rigid pass
(``tile_and_correct_3d`` ``:1518-1608`` with the class default ``shifts_opencv=True``, ``upsample_factor_fft=10`` as
``tile_and_correct_wrapper`` ``:2029-2037`` hard-codes it).  The registration runs on the GPU (K8 ``dnmf_register_patches``:
matrix-multiply DFTs, no FFT library; ``csrc/register_patches.hip``).

Not offered (``NotImplementedError``): rigid-only correction of the frames, the corrected movie, 2-D (cv2) registration,
``shifts_opencv=False`` (cubic resize of the shift field), memory-mapped files, ``dview``.  ``template=None`` takes the
temporal median of the video instead of the reference's rigid pre-pass (``:298-301``).

Parity: the reference module cannot be imported in the build container (cv2, skimage and ``past`` are absent, ``np.int``
is gone from numpy 2) and ships no fixture; this class is checked against ``oracle/motion_oracle.py``, a numpy restatement
of the same functions -- **parity unpinned**.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import ops

device = 'cuda'


class MotionCorrect(object):
    def __init__(self, video, min_mov=None, dview=None, max_shifts=(6, 6, 1), niter_rig=1, splits_rig=1,
                 num_splits_to_process_rig=None, strides=(96, 96, 1), overlaps=(32, 32, 1), splits_els=1,
                 num_splits_to_process_els=None, upsample_factor_grid=4, max_deviation_rigid=3, shifts_opencv=True,
                 nonneg_movie=True, gSig_filt=None, use_cuda=False, border_nan=True, pw_rigid=False, num_frames_split=80,
                 var_name_hdf5='mov', is3D=True, indices=(slice(None), slice(None))):
        ops._lib.load()   # fail here, loudly, if the HIP library is not built
        if not is3D:
            raise NotImplementedError("MotionCorrect: only the 3-D functions are built (apply_shifts_points is 3-D)")
        if not shifts_opencv:
            raise NotImplementedError("MotionCorrect: shifts_opencv=False (cubic resize of the shift field) is not built")
        if gSig_filt is not None or dview is not None:
            raise NotImplementedError("MotionCorrect: gSig_filt / dview are not built")
        if type(video) is not list:
            video = [video]
        self.video = video
        self.max_shifts = tuple(int(v) for v in max_shifts)
        self.strides = tuple(int(v) for v in strides)
        self.overlaps = tuple(int(v) for v in overlaps)
        if not (len(self.max_shifts) == len(self.strides) == len(self.overlaps) == 3):
            raise ValueError("MotionCorrect: max_shifts, strides and overlaps need three entries (x, y, z)")
        self.max_deviation_rigid = int(max_deviation_rigid)
        self.upsample_factor_grid = upsample_factor_grid
        self.shifts_opencv = True
        self.min_mov = min_mov
        self.nonneg_movie = nonneg_movie
        self.border_nan = border_nan
        self.pw_rigid = bool(pw_rigid)
        self.is3D = True
        self.upsample_factor_fft = 10     # tile_and_correct_wrapper :2029-2037

    @staticmethod
    def _frames(video):
        """(T, X, Y, Z) numpy / torch -> (T, P) fp32 CUDA rows and the volume size."""
        v = torch.as_tensor(np.asarray(video) if not torch.is_tensor(video) else video)
        if v.dim() != 4:
            raise ValueError(f"MotionCorrect: a video is (T, X, Y, Z), got {tuple(v.shape)}")
        sz = [int(s) for s in v.shape[1:]]
        return v.to(device, torch.float32).reshape(v.shape[0], -1).contiguous(), sz

    def motion_correct(self, template=None):
        """Reference :175-211; only the piecewise-rigid shifts are built."""
        if not self.pw_rigid:
            raise NotImplementedError("MotionCorrect.motion_correct: rigid correction of the frames is not built; "
                                      "set pw_rigid=True (the shifts apply_shifts_points needs)")
        self.motion_correct_pwrigid(template=template)
        b0 = np.ceil(np.max([np.max(np.abs(self.x_shifts_els)), np.max(np.abs(self.y_shifts_els)),
                             np.max(np.abs(self.z_shifts_els))]))
        self.border_to_0 = int(b0)
        return self

    def motion_correct_pwrigid(self, template=None, show_template=False):
        """Reference :260-328: fills ``x_shifts_els``, ``y_shifts_els``, ``z_shifts_els`` (one (NP,) array per frame),
        ``shifts_rig`` (the rigid shift of every frame), ``coord_shifts_els`` (the patch grid indices) and
        ``total_template_els``."""
        self.x_shifts_els, self.y_shifts_els, self.z_shifts_els = [], [], []
        self.coord_shifts_els, self.shifts_rig = [], []
        for video_cur in self.video:
            frames, sz = self._frames(video_cur)
            if self.min_mov is None:
                self.min_mov = float(frames.min())           # :196-199
            if template is None:
                tmpl = frames.median(0).values               # (the reference: a rigid pre-pass, :298-301)
            else:
                tmpl = torch.as_tensor(np.asarray(template) if not torch.is_tensor(template) else template).to(
                    device, torch.float32).reshape(-1)
            self.total_template_els = tmpl.view(*sz)
            rigid, patch = ops.register_patches(frames, tmpl, sz, self.strides, self.overlaps, self.max_shifts,
                                                self.max_deviation_rigid, self.upsample_factor_fft,
                                                add_to_movie=-self.min_mov)
            dims, starts = ops.patch_grid(sz, self.strides, self.overlaps)
            grid = [tuple(int(v) for v in np.unravel_index(q, dims)) for q in range(len(starts))]
            p = patch.cpu().numpy()
            for t in range(p.shape[0]):
                self.x_shifts_els.append(p[t, :, 0].copy())
                self.y_shifts_els.append(p[t, :, 1].copy())
                self.z_shifts_els.append(p[t, :, 2].copy())
                self.coord_shifts_els.append(grid)
            self.shifts_rig += [tuple(r) for r in rigid.cpu().numpy()]
            self._patch_shifts = patch                       # (T, NP, 3) on the GPU, for apply_shifts_points

    def _centers(self, sz):
        _, starts = ops.patch_grid(sz, self.strides, self.overlaps)
        return starts.astype(np.float64) + np.array(self.strides, dtype=np.float64) / 2     # :366

    def _shift_table(self, T):
        """(T, NP, 3) fp32 CUDA from the lists (a caller may have edited them)."""
        sh = np.stack([np.stack(self.x_shifts_els[:T]), np.stack(self.y_shifts_els[:T]), np.stack(self.z_shifts_els[:T])], 2)
        return torch.from_numpy(sh.astype(np.float32)).to(device)

    def apply_shifts_points(self, video, points):
        """Reference :351-371: ``P_T`` (K, 3, T) float64 numpy -- point k in frame t, moved by the shifts of the patch whose
        centre is nearest to it, relative to frame 0."""
        v = np.asarray(video.shape) if hasattr(video, "shape") else None
        T, sz = int(v[0]), [int(s) for s in v[1:]]
        pts = torch.as_tensor(np.asarray(points)).to(device, torch.float32).contiguous()
        centers = torch.from_numpy(self._centers(sz)).to(device, torch.float32).contiguous()
        out = ops.apply_shifts_points(pts, self._shift_table(T), centers)
        return out.double().cpu().numpy()

    def apply_shifts_frame(self, video, points, t):
        """Reference :330-349: the points moved by frame t's shifts (no reference frame, all three signs +)."""
        from scipy.spatial import distance
        v = np.asarray(video.shape)
        sz = [int(s) for s in v[1:]]
        pts = np.asarray(points, dtype=np.float64)
        idx = distance.cdist(self._centers(sz), pts).argmin(0)
        A = pts.copy()
        A[:, 0] += np.asarray(self.x_shifts_els[t], dtype=np.float32)[idx]
        A[:, 1] += np.asarray(self.y_shifts_els[t], dtype=np.float32)[idx]
        A[:, 2] += np.asarray(self.z_shifts_els[t], dtype=np.float32)[idx]
        return A

    def get_params(self):
        return {'max_shifts': self.max_shifts, 'strides': self.strides, 'overlaps': self.overlaps,
                'upsample_factor_grid': self.upsample_factor_grid, 'max_deviation_rigid': self.max_deviation_rigid,
                'shifts_opencv': self.shifts_opencv, 'nonneg_movie': self.nonneg_movie, 'border_nan': self.border_nan,
                'is3D': self.is3D}
