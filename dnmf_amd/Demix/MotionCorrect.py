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

``template=None`` runs the reference's rigid pre-pass first (``motion_correct_rigid`` ``:213-258`` ->
``motion_correct_batch_rigid`` ``:1770-1877``): the binned median of the video as the first template, every frame
registered to it and moved by its shift through the phases of its spectrum (``apply_shifts_dft``), the NaN-aware mean of the
moved frames as the template of the piecewise pass -- K8 ``dnmf_rigid_correct``.  ``motion_correct_rigid`` is also offered
on its own (``shifts_rig``, ``total_template_rig``, ``templates_rig``; the corrected movie in ``mc`` only with
``save_corrected=True``: it is as large as the video).

2-D videos (``is3D=False``, (T, X, Y); ``register_translation`` ``:801-1024``, ``tile_and_correct`` ``:1272-1418``) are
registered as one slice of the same kernels: ``motion_correct_pwrigid(template=...)`` fills ``x_shifts_els`` /
``y_shifts_els``.  (``apply_shifts_points`` is 3-D in the reference too.)

Not offered (``NotImplementedError``): the piecewise-corrected movie (``cv2.remap``), the 2-D rigid correction
(``cv2.warpAffine``), ``shifts_opencv=False`` (cubic resize of the shift field), memory-mapped files, ``dview``, ``gSig_filt``.

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
                 var_name_hdf5='mov', is3D=True, indices=(slice(None), slice(None)), save_corrected=False):
        ops._lib.load()   # fail here, loudly, if the HIP library is not built
        if not shifts_opencv:
            raise NotImplementedError("MotionCorrect: shifts_opencv=False (cubic resize of the shift field) is not built")
        if gSig_filt is not None or dview is not None:
            raise NotImplementedError("MotionCorrect: gSig_filt / dview are not built")
        if type(video) is not list:
            video = [video]
        self.video = video
        self.is3D = bool(is3D)
        self.max_shifts = tuple(int(v) for v in max_shifts)
        self.strides = tuple(int(v) for v in strides)
        self.overlaps = tuple(int(v) for v in overlaps)
        nd = 3 if self.is3D else 2
        if not self.is3D:
            # 2-D videos (T, X, Y) -- register_translation :801-1024, tile_and_correct :1272-1418 -- run as one slice of the
            # 3-D kernels: a third axis of one voxel has no window and a zero shift.  (A 2-D call may still pass the 3-entry
            # defaults of this class: their third entries are dropped.)
            self.max_shifts, self.strides, self.overlaps = self.max_shifts[:2], self.strides[:2], self.overlaps[:2]
        if not (len(self.max_shifts) == len(self.strides) == len(self.overlaps) == nd):
            raise ValueError(f"MotionCorrect: max_shifts, strides and overlaps need {nd} entries")
        self.max_deviation_rigid = int(max_deviation_rigid)
        self.upsample_factor_grid = upsample_factor_grid
        self.shifts_opencv = True
        self.min_mov = min_mov
        self.nonneg_movie = nonneg_movie
        if border_nan not in (True, False, 'min', 'copy'):
            raise ValueError(f"MotionCorrect: border_nan {border_nan!r}")
        self.border_nan = border_nan
        self.niter_rig = int(niter_rig)
        if splits_rig != 1 or splits_els != 1 or num_splits_to_process_rig is not None or num_splits_to_process_els is not None:
            raise NotImplementedError("MotionCorrect: splits (chunks for a process pool) are not built; the GPU walks the video "
                                      "in pieces of its own")
        self.save_corrected = bool(save_corrected)
        self.pw_rigid = bool(pw_rigid)
        self.upsample_factor_fft = 10     # tile_and_correct_wrapper :2029-2037

    def _p3(self):
        """strides, overlaps, max_shifts with three entries (one slice for a 2-D video: window 1, no shift)."""
        if self.is3D:
            return self.strides, self.overlaps, self.max_shifts
        return self.strides + (1,), self.overlaps + (0,), self.max_shifts + (0,)

    def _frames(self, video):
        """(T, X, Y, Z) -- (T, X, Y) for is3D=False -- numpy / torch -> (T, P) fp32 CUDA rows and the volume size (Z = 1)."""
        v = torch.as_tensor(np.asarray(video) if not torch.is_tensor(video) else video)
        if not self.is3D and v.dim() == 3:
            v = v[..., None]
        if v.dim() != 4 or (not self.is3D and v.shape[3] != 1):
            raise ValueError(f"MotionCorrect: a video is (T, X, Y, Z) (is3D) or (T, X, Y), got {tuple(v.shape)}")
        sz = [int(s) for s in v.shape[1:]]
        return v.to(device, torch.float32).reshape(v.shape[0], -1).contiguous(), sz

    def motion_correct(self, template=None):
        """Reference :176-211."""
        if self.min_mov is None:
            self.min_mov = min(float(torch.as_tensor(v).min()) for v in self.video)     # (:193-195: of the first video)
        if self.pw_rigid:
            self.motion_correct_pwrigid(template=template)
            b0 = np.ceil(np.max([np.max(np.abs(self.x_shifts_els)), np.max(np.abs(self.y_shifts_els))] +
                                ([np.max(np.abs(self.z_shifts_els))] if self.is3D else [])))
        else:
            self.motion_correct_rigid(template=template)
            b0 = np.ceil(np.max(np.abs(self.shifts_rig)))
        self.border_to_0 = int(b0)
        return self

    @staticmethod
    def _bin_median_3d(frames, window=10):
        """bin_median_3d :464-494 on (T, P) rows: mean over groups of frames (group j = frames j, j + num_windows, ...: the
        reference's reshape), NaN-aware median over the groups (mean of the two middle values for an even count)."""
        T = frames.shape[0]
        window = min(window, T)
        nw = T // window
        bins = frames[:nw * window].view(window, nw, -1).nanmean(0)          # (nw, P)
        srt = bins.sort(0).values                                            # NaNs last
        cnt = (~torch.isnan(bins)).sum(0).clamp(min=1)
        lo = srt.gather(0, ((cnt - 1) // 2)[None])[0]
        hi = srt.gather(0, (cnt // 2)[None])[0]
        return 0.5 * (lo + hi)

    def motion_correct_rigid(self, template=None):
        """Reference :213-258 -> motion_correct_batch_rigid :1770-1877 (3-D, one chunk): fills ``total_template_rig``,
        ``templates_rig``, ``shifts_rig`` (one (x, y, z) tuple per frame, the registration's shift with its sign flipped,
        :1574) and -- only with ``save_corrected=True`` -- ``mc`` (one (X, Y, Z, T) array per video)."""
        if not self.is3D:
            # (the reference moves 2-D frames with cv2.warpAffine when shifts_opencv is set, :1349-1354, and its own 2-D start
            # without a template calls a method numpy arrays do not have, :1828)
            raise NotImplementedError("MotionCorrect.motion_correct_rigid: the 2-D rigid correction (cv2.warpAffine) is not built; "
                                      "2-D videos: motion_correct_pwrigid(template=...)")
        self.total_template_rig = template
        self.templates_rig, self.shifts_rig, self.mc = [], [], []
        for video_cur in self.video:
            frames, sz = self._frames(video_cur)
            if self.min_mov is None:
                self.min_mov = float(frames.min())
            if self.total_template_rig is None:
                tmpl = self._bin_median_3d(frames)                           # :1826
            else:
                tmpl = torch.as_tensor(np.asarray(self.total_template_rig) if not torch.is_tensor(self.total_template_rig)
                                       else self.total_template_rig).to(device, torch.float32).reshape(-1)
            add = float(np.float32(-self.min_mov))                           # (:2122: passed on as a float32)
            step = max(1, min(frames.shape[0], (1 << 30) // (4 * frames.shape[1]))) if self.save_corrected else frames.shape[0]
            rigid = None
            for it in range(max(1, self.niter_rig)):
                last = it == max(1, self.niter_rig) - 1
                tsum = tcount = None
                parts, moved = [], []
                for f0 in range(0, frames.shape[0], step):
                    r, out, tsum, tcount = ops.rigid_correct(frames[f0:f0 + step], tmpl, sz, self.max_shifts, self.upsample_factor_fft,
                                                             add_to_movie=add, border_nan=self.border_nan,
                                                             want_frames=self.save_corrected and last, tsum=tsum, tcount=tcount)
                    parts.append(r)
                    if out is not None:
                        moved.append(out.cpu())
                rigid = torch.cat(parts)
                new_temp = tsum / tcount                                     # nanmean :2057 (0 / 0: NaN)
                new_temp = torch.where(torch.isnan(new_temp), new_temp[~torch.isnan(new_temp)].min(), new_temp)   # :2058
                tmpl = new_temp
            if template is None:
                self.total_template_rig = tmpl.view(*sz).cpu().numpy()
            self.templates_rig.append(tmpl.view(*sz).cpu().numpy())
            self.shifts_rig += [tuple(float(-v) for v in row) for row in rigid.cpu().numpy()]
            if moved:
                self.mc.append(torch.cat(moved).view(-1, *sz).permute(1, 2, 3, 0).numpy())

    def motion_correct_pwrigid(self, template=None, show_template=False):
        """Reference :260-328: fills ``x_shifts_els``, ``y_shifts_els``, ``z_shifts_els`` (one (NP,) array per frame),
        ``shifts_rig`` (the rigid shift of every frame), ``coord_shifts_els`` (the patch grid indices) and
        ``total_template_els``."""
        self.x_shifts_els, self.y_shifts_els = [], []
        if self.is3D:
            self.z_shifts_els = []
        self.coord_shifts_els = []
        strides, overlaps, max_shifts = self._p3()
        for video_cur in self.video:
            frames, sz = self._frames(video_cur)
            if self.min_mov is None:
                self.min_mov = float(frames.min())           # :196-199
            if template is None:
                self.motion_correct_rigid()                  # :298-301 (every video of the list, as the reference does)
                tmpl = torch.from_numpy(np.ascontiguousarray(self.total_template_rig)).to(device, torch.float32).reshape(-1)
            else:
                tmpl = torch.as_tensor(np.asarray(template) if not torch.is_tensor(template) else template).to(
                    device, torch.float32).reshape(-1)
            self.total_template_els = tmpl.view(*sz) if self.is3D else tmpl.view(*sz[:2])
            rigid, patch = ops.register_patches(frames, tmpl, sz, strides, overlaps, max_shifts,
                                                self.max_deviation_rigid, self.upsample_factor_fft,
                                                add_to_movie=-self.min_mov)
            dims, starts = ops.patch_grid(sz, strides, overlaps)
            nd = 3 if self.is3D else 2
            grid = [tuple(int(v) for v in np.unravel_index(q, dims))[:nd] for q in range(len(starts))]
            p = patch.cpu().numpy()
            for t in range(p.shape[0]):
                self.x_shifts_els.append(p[t, :, 0].copy())
                self.y_shifts_els.append(p[t, :, 1].copy())
                if self.is3D:
                    self.z_shifts_els.append(p[t, :, 2].copy())
                self.coord_shifts_els.append(grid)
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
        if not self.is3D:
            raise NotImplementedError("MotionCorrect.apply_shifts_points is a 3-D function (reference :351-371: z_shifts_els, "
                                      "sliding_window_3d); register 2-D videos as (T, X, Y, 1) with is3D=True")
        v = np.asarray(video.shape) if hasattr(video, "shape") else None
        T, sz = int(v[0]), [int(s) for s in v[1:]]
        pts = torch.as_tensor(np.asarray(points)).to(device, torch.float32).contiguous()
        centers = torch.from_numpy(self._centers(sz)).to(device, torch.float32).contiguous()
        out = ops.apply_shifts_points(pts, self._shift_table(T), centers)
        return out.double().cpu().numpy()

    def apply_shifts_frame(self, video, points, t):
        """Reference :330-349: the points moved by frame t's shifts (no reference frame, all three signs +)."""
        if not self.is3D:
            raise NotImplementedError("MotionCorrect.apply_shifts_frame is a 3-D function (reference :330-349)")
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
