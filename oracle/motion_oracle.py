"""CPU restatement of the position initialiser of the reference: piecewise-rigid registration shifts of a 3-D video and
their application to neuron centres.  TEST INFRASTRUCTURE ONLY -- nothing under dnmf_amd/ imports this.

Reference: /root/reference/Demix/MotionCorrect.py (a vendored copy of CaImAn / NoRMCorre, called by nothing in the
reference tree).  **Parity unpinned**: that module cannot be imported in the build container (it needs cv2, skimage and
`past`, none of which is installed, and uses `np.int`, removed from numpy 2), and the reference ships no test, fixture or
output for it.  What follows restates, from the source text, exactly the part `apply_shifts_points` needs -- the shifts
`x_shifts_els / y_shifts_els / z_shifts_els` of the 3-D piecewise-rigid pass with the class defaults
(`shifts_opencv=True`) -- with `numpy.fft` where the reference calls `np.fft` (the 3-D functions never use cv2.dft):

  sliding_window_3d        MotionCorrect.py:1190-1221   the patch grid
  _upsampled_dft           :498-614                     matrix-multiply DFT of a small upsampled region
  register_translation_3d  :648-797                     integer peak of the circular cross-correlation inside a window,
                                                        refined to 1/upsample_factor by the upsampled DFT
  tile_shifts_3d           :1518-1608 (tile_and_correct_3d up to `total_shifts`), called per frame by
                           tile_and_correct_wrapper :2004-2060 with upsample_factor_fft=10
  apply_shifts_points      :351-371

  bin_median_3d            :464-494                     the first template
  apply_shifts_dft         :1028-1157 (3-D branch)      a frame moved by its rigid shift through its spectrum
  rigid_correct_3d         tile_and_correct_3d :1518-1574 with max_deviation_rigid == 0
  rigid_template           motion_correct_batch_rigid :1770-1877: the template a piecewise-rigid pass starts from when
                           the caller gives none (MotionCorrect.motion_correct_pwrigid :298-301), and shifts_rig

  register_translation     :801-1024 (2-D)              as the 3-D restatement on an (X, Y, 1) array
  tile_shifts_2d           tile_and_correct :1272-1418 up to `total_shifts`, sliding_window :1160-1188

Not restated (not needed for the shifts): the piecewise-corrected frames (`warp_sk`, cv2.remap), the rigidly corrected 2-D
frames (cv2.warpAffine), the
`shifts_opencv=False` branch (cubic resize of the shift field, skimage).
"""
from __future__ import annotations

import numpy as np


def patch_starts(size, overlap, stride):
    """Start offsets of the patches along one axis (sliding_window_3d :1207-1213): windows of overlap + stride voxels every
    `stride`, the last one flush with the end."""
    w = overlap + stride
    return list(range(0, size - w, stride)) + [size - w]


def sliding_window_3d(shape, overlaps, strides):
    """[(dim_1, dim_2, dim_3, x, y, z)] in the reference's iteration order (x outermost), :1214-1221."""
    r = [patch_starts(shape[d], overlaps[d], strides[d]) for d in range(3)]
    return [(i, j, k, x, y, z) for i, x in enumerate(r[0]) for j, y in enumerate(r[1]) for k, z in enumerate(r[2])]


def _signed_freq(n):
    """ifftshift(arange(n)) - floor(n / 2) (:585-598): the signed frequency of DFT index j."""
    return np.fft.ifftshift(np.arange(n)) - np.floor(n / 2)


def upsampled_dft(data, region, upsample_factor, axis_offsets):
    """_upsampled_dft :498-614 for 3-D data: sum_j data[j] exp(-2 pi i f_j (u - offset) / (n upsample_factor)) on a
    `region`^3 grid of u."""
    out = data
    kern = []
    for d in range(3):
        n = data.shape[d]
        u = np.arange(region) - axis_offsets[d]
        kern.append(np.exp((-2j * np.pi / (n * upsample_factor)) * np.outer(u, _signed_freq(n))))   # (region, n)
    out = np.tensordot(kern[0], out, axes=[1, 0])           # (r, n1, n2)
    out = np.tensordot(out, kern[1].T, axes=[1, 0])          # (r, n2, r)   :606
    out = np.tensordot(out, kern[2], axes=[1, 1])            # (r, r, r)    :611
    return out


def _zero_outside(cc, shifts_lb, shifts_ub, max_shifts):
    """The window of admissible shifts, by numpy slicing exactly as :727-747 writes it."""
    if shifts_lb is not None or shifts_ub is not None:
        for d in range(3):
            sl = [slice(None)] * 3
            if shifts_lb[d] < 0 and shifts_ub[d] >= 0:
                sl[d] = slice(shifts_ub[d], shifts_lb[d])
                cc[tuple(sl)] = 0
            else:
                sl[d] = slice(None, shifts_lb[d])
                cc[tuple(sl)] = 0
                sl[d] = slice(shifts_ub[d], None)
                cc[tuple(sl)] = 0
    else:
        for d in range(3):
            sl = [slice(None)] * 3
            sl[d] = slice(max_shifts[d], -max_shifts[d])
            cc[tuple(sl)] = 0
    return cc


def register_translation_3d(src_image, target_image, upsample_factor=1, shifts_lb=None, shifts_ub=None, max_shifts=(10, 10, 10),
                            full_output=False, to_complex64=True):
    """:648-797, space='real'.  Returns the shift vector (3,) float64; with full_output also the spectrum of the source and
    the phase difference, as the reference returns them (:797).  The images are rounded to complex64 first (:712-715); the
    transforms are then taken in double precision, as the numpy of the reference's time did for any input."""
    first = np.complex64 if to_complex64 else np.complex128
    src_freq = np.fft.fftn(np.array(src_image, dtype=first).astype(np.complex128))
    target_freq = np.fft.fftn(np.array(target_image, dtype=first).astype(np.complex128))
    shape = src_freq.shape
    image_product = src_freq * target_freq.conj()
    cross_correlation = np.fft.ifftn(image_product)
    CCmax = cross_correlation.max()          # (numpy's max of a complex array: by real part, then imaginary, :727)
    new_cc = _zero_outside(np.abs(cross_correlation), shifts_lb, shifts_ub, max_shifts)
    maxima = np.unravel_index(np.argmax(new_cc), new_cc.shape)
    midpoints = np.array([np.fix(s // 2) for s in shape])
    shifts = np.array(maxima, dtype=np.float32)
    shifts[shifts > midpoints] -= np.array(shape)[shifts > midpoints]
    if upsample_factor > 1:
        shifts = np.round(shifts * upsample_factor) / upsample_factor
        region = int(np.ceil(upsample_factor * 1.5))
        dftshift = np.fix(region / 2.0)
        offset = dftshift - shifts * upsample_factor
        cc = upsampled_dft(image_product.conj(), region, float(upsample_factor), offset).conj()
        cc = cc / (src_freq.size * float(upsample_factor) ** 2)
        maxima = np.array(np.unravel_index(np.argmax(np.abs(cc)), cc.shape), dtype=np.float64) - dftshift
        shifts = shifts + maxima / upsample_factor
        CCmax = cc.max()
    for d in range(3):
        if shape[d] == 1:
            shifts[d] = 0
    if full_output:
        return np.asarray(shifts, dtype=np.float64), src_freq, float(np.arctan2(CCmax.imag, CCmax.real))
    return np.asarray(shifts, dtype=np.float64)


def apply_shifts_dft_3d(src_freq, shifts, diffphase, border_nan=True):
    """apply_shifts_dft :1028-1157 for a 3-D spectrum: the image moved by `shifts` through the phases of its spectrum, real
    part, and the border where the shift brought in voxels from the other side: NaN (True), the image's smallest value
    ('min'), the nearest row / column / slice inside ('copy'), or left alone (False).

    What is kept from the reference.  It swaps the first two shifts (:1083) and pairs the swapped shifts[0] with the
    frequencies of axis 1 and shifts[1] with those of axis 0 (:1085-1091): every axis is ramped by its own shift, the phase
    terms added in the order axis 1, axis 0, axis 2.  The borders (:1098-1145) then use the SWAPPED pair on the axes in
    order: the border of axis 0 follows the shift of axis 1 and the other way round; axis 2 its own."""
    own = [float(shifts[0]), float(shifts[1]), float(shifts[2])]
    n = src_freq.shape
    freq = [np.fft.ifftshift(np.arange(-np.fix(n[d] / 2.), np.ceil(n[d] / 2.))).reshape([-1 if e == d else 1 for e in range(3)])
            for d in range(3)]
    turns = None
    for d in (1, 0, 2):
        term = own[d] * freq[d] / float(n[d])
        turns = term if turns is None else turns + term
    moved = src_freq * np.exp(1j * 2 * np.pi * turns)
    moved = moved * np.exp(1j * diffphase)
    out = np.real(np.fft.ifftn(moved))
    if border_nan is False:
        return out
    swapped = [own[1], own[0], own[2]]
    fill = np.nanmin(out) if border_nan == 'min' else np.nan
    if border_nan not in (True, 'min', 'copy'):
        raise ValueError(border_nan)
    for d in range(3):
        head = int(np.ceil(max(0.0, swapped[d])))            # rows [:head] ...
        tail = int(np.floor(min(0.0, swapped[d])))           # ... and, when negative, rows [tail:]
        view = np.moveaxis(out, d, 0)
        if border_nan == 'copy':
            if head > 0 or d == 0:                           # (:1133 copies into axis 0 unconditionally: a no-op for head == 0)
                view[:head] = view[head]
            if tail < 0:
                view[tail:] = view[tail - 1]
        else:
            view[:head] = fill
            if tail < 0:
                view[tail:] = fill
    return out


def bin_median_3d(mat, window=10):
    """:464-494 (exclude_nans=True): mean over groups of frames, median over the groups.  (The reshape to (window,
    num_windows, ...) makes group j the frames j, j + num_windows, ...; kept.)"""
    T, d1, d2, d3 = mat.shape
    if T < window:
        window = T
    num_windows = int(T // window)
    num_frames = num_windows * window
    return np.nanmedian(np.nanmean(np.reshape(mat[:num_frames], (window, num_windows, d1, d2, d3)), axis=0), axis=0)


def rigid_correct_3d(img, template, max_shifts, upsample_factor_fft=10, add_to_movie=0.0, border_nan=True):
    """tile_and_correct_3d :1518-1574 with max_deviation_rigid == 0: (corrected frame, total shift) -- the frame moved
    by its rigid shift, the shift with the sign flipped (:1573-1574)."""
    img = np.asarray(img, dtype=np.float64) + add_to_movie
    template = np.asarray(template, dtype=np.float64) + add_to_movie
    s, sfr_freq, diffphase = register_translation_3d(img, template, upsample_factor=upsample_factor_fft, max_shifts=max_shifts,
                                                     full_output=True)
    new_img = apply_shifts_dft_3d(sfr_freq, (s[0], s[1], s[2]), diffphase, border_nan=border_nan)
    return new_img - add_to_movie, (-s[0], -s[1], -s[2])


def rigid_template(video, max_shifts, num_iter=1, template=None, add_to_movie=None, upsample_factor_fft=10, border_nan=True):
    """motion_correct_batch_rigid :1770-1877 for a 3-D video (T, X, Y, Z) with splits = 1 (one chunk: every frame;
    tile_and_correct_wrapper :2004-2060 walks the whole video whatever the chunk's indices are): (total_template, shifts
    (T,3) with the flipped sign, corrected frames (T,X,Y,Z) float32)."""
    video = np.asarray(video)
    if template is None:
        template = bin_median_3d(video)
    if add_to_movie is None:
        add_to_movie = -np.min(template)
    add_to_movie = float(np.array(add_to_movie, dtype=np.float32))        # motion_correction_piecewise :2122
    new_templ = template
    for _ in range(num_iter):
        old_templ = new_templ.copy()
        mc = np.zeros(video.shape, dtype=np.float32)                       # :2027
        shifts = []
        for count, img in enumerate(video):
            mc[count], sh = rigid_correct_3d(img, old_templ, max_shifts, upsample_factor_fft, add_to_movie, border_nan)
            shifts.append(sh)
        new_temp = np.nanmean(mc, 0)
        new_temp[np.isnan(new_temp)] = np.nanmin(new_temp)
        new_templ = np.nanmedian(np.stack([new_temp]), 0)
    return new_templ, np.array(shifts, dtype=np.float64), mc


def register_translation(src_image, target_image, upsample_factor=1, shifts_lb=None, shifts_ub=None, max_shifts=(10, 10)):
    """The 2-D register_translation :801-1024 (space='real'): the same algorithm on two axes -- window, argmax, upsampled
    DFT -- so it is evaluated here as the 3-D restatement on an (X, Y, 1) array: a third axis of one voxel leaves every
    window of :727-747 empty and its shift zero (:789-791).  Differences kept: the 2-D function takes the images to
    complex128 directly (no complex64 step, :919-924); it calls cv2.dft where this uses numpy.fft (cv2 is absent; the forward
    scale factor of cv2.DFT_SCALE does not move a maximum)."""
    ext = lambda v, z: None if v is None else np.array(list(v) + [z])      # (z window [-1, 1): zeroes nothing of one slice)
    s = register_translation_3d(np.asarray(src_image)[:, :, None], np.asarray(target_image)[:, :, None], upsample_factor,
                                ext(shifts_lb, -1), ext(shifts_ub, 1), tuple(max_shifts) + (0,), to_complex64=False)
    return s[:2]


def tile_shifts_2d(img, template, strides, overlaps, max_shifts, upsample_factor_fft=10, max_deviation_rigid=3, add_to_movie=0.0):
    """tile_and_correct :1272-1418 up to its `total_shifts` (shifts_opencv=True branch): (rigid (2,), total_shifts (NP,2) =
    (-x, -y) :1415-1416) over the patches of sliding_window :1160-1188."""
    img = np.asarray(img, dtype=np.float64) + add_to_movie
    template = np.asarray(template, dtype=np.float64) + add_to_movie
    rigid = register_translation(img, template, upsample_factor_fft, max_shifts=max_shifts)
    lb = np.ceil(np.subtract(rigid, max_deviation_rigid)).astype(int)
    ub = np.floor(np.add(rigid, max_deviation_rigid)).astype(int)
    w = np.add(overlaps, strides)
    out = []
    for x in patch_starts(img.shape[0], overlaps[0], strides[0]):
        for y in patch_starts(img.shape[1], overlaps[1], strides[1]):
            sh = register_translation(img[x:x + w[0], y:y + w[1]], template[x:x + w[0], y:y + w[1]], upsample_factor_fft,
                                      shifts_lb=lb, shifts_ub=ub, max_shifts=max_shifts)
            out.append((-sh[0], -sh[1]))
    return rigid, np.array(out, dtype=np.float64)


def tile_shifts_3d(img, template, strides, overlaps, max_shifts, upsample_factor_fft=10, max_deviation_rigid=3,
                   add_to_movie=0.0):
    """tile_and_correct_3d :1518-1608 up to its `total_shifts` (shifts_opencv=True branch): the rigid shift, then one
    shift per patch inside [rigid - max_deviation, rigid + max_deviation].  Returns (rigid (3,), total_shifts (NP,3)) with
    the reference's sign convention (-x, -y, +z) :1596-1597."""
    img = np.asarray(img, dtype=np.float64) + add_to_movie
    template = np.asarray(template, dtype=np.float64) + add_to_movie
    rigid = register_translation_3d(img, template, upsample_factor=upsample_factor_fft, max_shifts=max_shifts)
    lb = np.ceil(np.subtract(rigid, max_deviation_rigid)).astype(int)
    ub = np.floor(np.add(rigid, max_deviation_rigid)).astype(int)
    w = np.add(overlaps, strides)
    out = []
    for (_, _, _, x, y, z) in sliding_window_3d(img.shape, overlaps, strides):
        a = img[x:x + w[0], y:y + w[1], z:z + w[2]]
        b = template[x:x + w[0], y:y + w[1], z:z + w[2]]
        s = register_translation_3d(a, b, upsample_factor_fft, shifts_lb=lb, shifts_ub=ub, max_shifts=max_shifts)
        out.append((-s[0], -s[1], s[2]))
    return rigid, np.array(out, dtype=np.float64)


def patch_centers(shape, overlaps, strides):
    """Patch start + strides / 2 (:366), (NP,3)."""
    return np.array([np.array(it[3:6]) + np.array(strides) / 2 for it in sliding_window_3d(shape, overlaps, strides)], dtype=np.float64)


def apply_shifts_points(shifts_x, shifts_y, shifts_z, shape, overlaps, strides, points):
    """MotionCorrect.apply_shifts_points :351-371.  shifts_* (T, NP); points (K,3).  Returns P_T (K,3,T) float64."""
    from scipy.spatial import distance
    points = np.asarray(points, dtype=np.float64)
    T = shifts_x.shape[0]
    idx = distance.cdist(patch_centers(shape, overlaps, strides), points).argmin(0)
    P_T = np.zeros((points.shape[0], 3, T))
    for t in range(T):
        P_T[:, :, t] = points
        P_T[:, 0, t] = P_T[:, 0, t] - shifts_x[t, idx] + shifts_x[0, idx]
        P_T[:, 1, t] = P_T[:, 1, t] - shifts_y[t, idx] + shifts_y[0, idx]
        P_T[:, 2, t] = P_T[:, 2, t] + shifts_z[t, idx] - shifts_z[0, idx]
    return P_T


def pw_rigid_shifts(video, template, strides, overlaps, max_shifts, upsample_factor_fft=10, max_deviation_rigid=3,
                    add_to_movie=0.0):
    """x/y/z_shifts_els of MotionCorrect.motion_correct_pwrigid (:260-328 -> motion_correct_batch_pwrigid :1880-2000 ->
    tile_and_correct_wrapper :2004-2060) for a video (T,X,Y,Z) against a given template: three (T, NP) arrays, plus the
    rigid shifts (T,3)."""
    sx, sy, sz, rig = [], [], [], []
    for img in video:
        r, ts = tile_shifts_3d(img, template, strides, overlaps, max_shifts, upsample_factor_fft, max_deviation_rigid, add_to_movie)
        rig.append(r)
        sx.append(ts[:, 0]), sy.append(ts[:, 1]), sz.append(ts[:, 2])
    return np.array(sx), np.array(sy), np.array(sz), np.array(rig)
