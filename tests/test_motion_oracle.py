"""oracle/motion_oracle.py (CPU restatement of the reference's Demix/MotionCorrect.py functions behind apply_shifts_points;
parity unpinned: the module cannot be imported here and has no fixtures) against properties the algorithm must have:
circular shifts by whole voxels and by tenths are recovered exactly, the patch grid is sliding_window_3d's, points follow
their nearest patch relative to frame 0."""
import numpy as np


def smooth_volume(shape, seed=0):
    from scipy.ndimage import gaussian_filter
    return gaussian_filter(np.random.RandomState(seed).rand(*shape), (2.0, 2.0, 0.8), mode="wrap")


def fourier_shift(v, d):
    """v moved by the (fractional) displacement d with periodic (Fourier) interpolation: out(x) = v(x - d)."""
    F = np.fft.fftn(v)
    for ax, n in enumerate(v.shape):
        f = np.fft.fftfreq(n)
        ph = np.exp(-2j * np.pi * f * d[ax])
        if n % 2 == 0:
            ph[n // 2] = np.cos(np.pi * d[ax])      # keep the result real
        F = F * ph.reshape([-1 if a == ax else 1 for a in range(3)])
    return np.fft.ifftn(F).real


def test_patch_grid_is_sliding_window_3d():
    from oracle import motion_oracle as MO
    assert MO.patch_starts(50, 8, 16) == [0, 16, 26]            # range(0, 50 - 24, 16) + [50 - 24]
    assert MO.patch_starts(24, 8, 16) == [0]
    g = MO.sliding_window_3d((50, 40, 2), (8, 8, 1), (16, 12, 1))
    assert g[0] == (0, 0, 0, 0, 0, 0) and g[1][:3] == (0, 1, 0) and g[-1][3:] == (26, 20, 0)
    np.testing.assert_array_equal(MO.patch_centers((50, 40, 2), (8, 8, 1), (16, 12, 1))[0], [8, 6, 0.5])


def test_register_translation_recovers_circular_shifts():
    from oracle import motion_oracle as MO
    tm = smooth_volume((40, 36, 4))
    img = np.roll(tm, (3, -2, 1), axis=(0, 1, 2))
    np.testing.assert_allclose(MO.register_translation_3d(img, tm, 1, max_shifts=(6, 6, 2)), [3, -2, 1])
    np.testing.assert_allclose(MO.register_translation_3d(img, tm, 10, max_shifts=(6, 6, 2)), [3, -2, 1], atol=1e-6)
    img = fourier_shift(tm, (2.3, -1.6, 0.0))
    np.testing.assert_allclose(MO.register_translation_3d(img, tm, 10, max_shifts=(6, 6, 2)), [2.3, -1.6, 0.0], atol=1e-6)
    # the window: a shift outside max_shifts is not found
    far = np.roll(tm, 9, axis=0)
    assert abs(MO.register_translation_3d(far, tm, 1, max_shifts=(6, 6, 2))[0]) < 6
    # the window around a rigid estimate: [ceil(r - dev), floor(r + dev))
    s = MO.register_translation_3d(img, tm, 10, shifts_lb=np.array([0, -4, -3]), shifts_ub=np.array([5, 2, 3]), max_shifts=(6, 6, 2))
    np.testing.assert_allclose(s, [2.3, -1.6, 0.0], atol=1e-6)


def test_tile_shifts_and_points():
    from oracle import motion_oracle as MO
    tm = smooth_volume((48, 40, 2), 1)
    video = np.array([tm, fourier_shift(tm, (1.2, 0.7, 0.0)), fourier_shift(tm, (-2.0, 1.5, 0.0))])
    sx, sy, sz, rig = MO.pw_rigid_shifts(video, tm, (16, 12, 1), (8, 8, 1), (5, 5, 1))
    np.testing.assert_allclose(rig[1], [1.2, 0.7, 0], atol=1e-6)
    assert sx.shape == (3, len(MO.sliding_window_3d((48, 40, 2), (8, 8, 1), (16, 12, 1))))
    # total_shifts = (-x, -y, +z); a small patch of a non-periodic field under-estimates its shift (circular correlation)
    assert -1.6 < np.median(sx[1]) < -0.4 and -1.9 < np.median(sy[2]) < -0.4
    pts = np.array([[10.0, 10.0, 0.5], [40.0, 30.0, 1.0]])
    P_T = MO.apply_shifts_points(sx, sy, sz, (48, 40, 2), (8, 8, 1), (16, 12, 1), pts)
    assert P_T.shape == (2, 3, 3)
    np.testing.assert_allclose(P_T[:, :, 0], pts)
    # a point follows the video: P_T = p - (shift[t] - shift[0]) with shift = -displacement
    assert 0.3 < P_T[0, 0, 1] - pts[0, 0] < 1.7 and 0.3 < P_T[1, 1, 2] - pts[1, 1] < 2.0


def test_rigid_pass_of_the_oracle():
    """apply_shifts_dft undoes the shift register_translation_3d finds (an integer shift: exactly a roll; the NaN borders as
    the reference places them -- the border of axis 0 from the shift of axis 1 and the other way round, on the side the
    2-D code would use, which is not where the roll wrapped around), the binned
    median follows the reference's reshape, and rigid_template returns a template every frame lines up with."""
    from oracle import motion_oracle as MO
    tm = smooth_volume((40, 36, 4), 2)
    img = np.roll(tm, (3, -2, 1), axis=(0, 1, 2))
    s, freq, phase = MO.register_translation_3d(img, tm, 10, max_shifts=(6, 6, 2), full_output=True)
    np.testing.assert_allclose(s, [3, -2, 1], atol=1e-6)
    assert abs(phase) < 1e-6
    moved = MO.apply_shifts_dft_3d(freq, s, phase, border_nan=False)
    np.testing.assert_allclose(moved, tm, atol=1e-5)   # new(r) = img(r + s): the displacement s is undone
    nan = np.isnan(MO.apply_shifts_dft_3d(freq, s, phase, border_nan=True))
    # shifts (3, -2, 1) -> swapped (-2, 3, 1): axis 0 loses its LAST two rows, axis 1 its FIRST three columns, axis 2 its first slice
    expect = np.zeros(tm.shape, bool)
    expect[-2:], expect[:, :3], expect[:, :, :1] = True, True, True
    np.testing.assert_array_equal(nan, expect)
    mat = np.arange(25 * 2 * 1 * 1, dtype=float).reshape(25, 2, 1, 1)
    # 25 frames: two bins; bin j = mean of frames j, j + 2, ..., j + 18; the median of two bins is their mean
    np.testing.assert_allclose(MO.bin_median_3d(mat)[:, 0, 0], [np.mean([mat[j::2][:10, i, 0, 0].mean() for j in range(2)]) for i in range(2)])
    video = np.array([fourier_shift(tm, sh) for sh in [(0, 0, 0), (1.2, 0.7, 0), (-2.0, 1.5, 0), (0.4, -0.6, 0)] * 3], dtype=np.float32)
    templ, shifts, mc = MO.rigid_template(video, (5, 5, 1))
    assert templ.shape == tm.shape and not np.isnan(templ).any() and mc.dtype == np.float32
    # every corrected frame registers to the new template with (almost) no shift left
    for f in mc[:4]:
        left = MO.register_translation_3d(np.where(np.isnan(f), templ, f), templ, 10, max_shifts=(5, 5, 1))
        assert np.abs(left).max() <= 0.2, left
    # total shifts differ between frames by minus the displacement differences
    np.testing.assert_allclose(shifts[1] - shifts[0], [-1.2, -0.7, 0], atol=0.11)


def test_border_modes_of_the_oracle():
    from oracle import motion_oracle as MO
    tm = smooth_volume((24, 20, 4), 3)
    img = np.roll(tm, (2, -3, 1), axis=(0, 1, 2))
    s, freq, phase = MO.register_translation_3d(img, tm, 10, max_shifts=(4, 4, 2), full_output=True)
    plain = MO.apply_shifts_dft_3d(freq, s, phase, border_nan=False)
    nan = np.isnan(MO.apply_shifts_dft_3d(freq, s, phase, border_nan=True))
    mn = MO.apply_shifts_dft_3d(freq, s, phase, border_nan='min')
    np.testing.assert_allclose(mn[nan], plain.min())
    np.testing.assert_allclose(mn[~nan], plain[~nan])
    cp = MO.apply_shifts_dft_3d(freq, s, phase, border_nan='copy')
    np.testing.assert_allclose(cp[~nan], plain[~nan])
    # swapped shifts (-3, 2, 1): axis 0 copies row X - 4 into its last three rows, axis 1 column 2 into its first two
    np.testing.assert_allclose(cp[-1, 5, 2], plain[-4, 5, 2])
    np.testing.assert_allclose(cp[6, 0, 2], plain[6, 2, 2])
    np.testing.assert_allclose(cp[6, 5, 0], plain[6, 5, 1])
    np.testing.assert_allclose(cp[-1, 0, 0], plain[-4, 2, 1])       # a corner: every index clamped


def test_two_dimensional_functions_of_the_oracle():
    from oracle import motion_oracle as MO
    tm = smooth_volume((48, 40, 1), 5)[:, :, 0]
    img = fourier_shift(tm[:, :, None], (2.3, -1.6, 0.0))[:, :, 0]
    np.testing.assert_allclose(MO.register_translation(img, tm, 10, max_shifts=(6, 6)), [2.3, -1.6], atol=1e-6)
    rigid, ts = MO.tile_shifts_2d(img, tm, (16, 12), (8, 8), (5, 5))
    np.testing.assert_allclose(rigid, [2.3, -1.6], atol=1e-6)
    assert ts.shape == (len(MO.patch_starts(48, 8, 16)) * len(MO.patch_starts(40, 8, 12)), 2)
    # (-x, -y); a small patch of a non-periodic field under-estimates its shift (circular correlation)
    assert -3.0 < np.median(ts[:, 0]) < -0.3 and 0.3 < np.median(ts[:, 1]) < 2.2
    # the same numbers as the 3-D functions on one slice
    r3, t3 = MO.tile_shifts_3d(img[:, :, None], tm[:, :, None], (16, 12, 1), (8, 8, 0), (5, 5, 0))
    np.testing.assert_allclose(ts, t3[:, :2], atol=0.1 + 1e-9)      # (complex64 rounding in the 3-D function only: a bin at most)
    assert (np.abs(ts - t3[:, :2]) < 1e-9).mean() > 0.9
