"""K8, the position initialiser: dnmf_register_patches / dnmf_apply_shifts_points and the
MotionCorrect-shaped front end against oracle/motion_oracle.py, a numpy restatement of the reference's
Demix/MotionCorrect.py (register_translation_3d :648-797, _upsampled_dft :498-614, sliding_window_3d :1190-1221,
tile_and_correct_3d :1518-1608, apply_shifts_points :351-371).

**Parity unpinned**: the reference module cannot be imported here (cv2 / skimage / past absent, np.int) and holds no
fixture; these tests pin the HIP path to the restatement only.  The shifts are quantised to 1 / upsample_factor = 0.1
voxel; the GPU works in fp32 where the restatement (like the reference) uses complex128, so a shift may land in the
neighbouring bin where two bins of the upsampled correlation tie to ~1e-6: the tests ask for identical shifts in at
least 97 % of the patches and never more than one bin apart.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def MO():
    from oracle import motion_oracle
    return motion_oracle


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dnmf_amd import ops
    return ops


def synthetic_video(sz, T, K, seed, piecewise=True):
    """Gaussian blobs on a dim background; frame t = the template moved by a smooth, piecewise different displacement
    (one shift per quadrant of the volume, blended) of up to ~3 voxels in x, y and a fraction of a slice in z."""
    rng = np.random.RandomState(seed)
    X, Y, Z = sz
    gx, gy, gz = np.meshgrid(np.arange(X), np.arange(Y), np.arange(Z), indexing="ij")
    pos = rng.rand(K, 3) * np.array([X, Y, Z])

    def render(shift_of):
        v = np.full(sz, 0.02)
        for k in range(K):
            d = shift_of(pos[k])
            v += np.exp(-(((gx - pos[k, 0] - d[0]) / 2.5) ** 2 + ((gy - pos[k, 1] - d[1]) / 2.5) ** 2 + ((gz - pos[k, 2] - d[2]) / 1.5) ** 2))
        return v

    template = render(lambda p: np.zeros(3))
    video, truth = [], []
    for t in range(T):
        base = rng.uniform(-2.5, 2.5, 3) * np.array([1, 1, 0.15])
        quad = rng.uniform(-1.0, 1.0, (2, 2, 3)) * np.array([1, 1, 0.0]) if piecewise else np.zeros((2, 2, 3))

        def shift_of(p, base=base, quad=quad):
            return base + quad[int(p[0] >= X / 2), int(p[1] >= Y / 2)]

        video.append(render(shift_of) + 0.002 * rng.randn(*sz))
        truth.append((base, quad))
    return np.array(video, dtype=np.float32), template.astype(np.float32), pos, truth


@pytest.mark.parametrize("sz,strides,overlaps,max_shifts", [
    ([48, 40, 2], (16, 12, 1), (8, 8, 1), (5, 5, 1)),       # two slices like the reference's demo volume
    ([40, 36, 5], (12, 12, 2), (8, 6, 1), (6, 4, 2)),       # odd sizes, several patch layers in z
    ([64, 64, 1], (24, 24, 1), (8, 8, 0), (6, 6, 0)),       # a single slice
])
def test_register_patches_vs_oracle(ops, MO, sz, strides, overlaps, max_shifts):
    T = 6
    video, template, _, _ = synthetic_video(sz, T, 40, seed=sum(sz))
    add = -float(video.min())
    sx, sy, sz_, rig = MO.pw_rigid_shifts(video, template, strides, overlaps, max_shifts, 10, 3, add)
    frames = torch.from_numpy(video.reshape(T, -1)).cuda()
    rigid, patch = ops.register_patches(frames, torch.from_numpy(template).cuda(), sz, strides, overlaps, max_shifts, 3, 10, add)
    dims, starts = ops.patch_grid(sz, strides, overlaps)
    ref_grid = MO.sliding_window_3d(sz, overlaps, strides)
    assert len(starts) == len(ref_grid) == sx.shape[1]
    np.testing.assert_array_equal(starts, np.array([g[3:6] for g in ref_grid]))
    np.testing.assert_array_equal(dims, np.array(ref_grid[-1][:3]) + 1)
    got = patch.cpu().numpy().astype(np.float64)
    ref = np.stack([sx, sy, sz_], 2)
    bins = np.abs(got - ref) * 10
    assert bins.max() <= 1.0 + 1e-3, bins.max()                       # never more than one bin of 0.1 voxel apart
    assert (bins < 1e-3).mean() >= 0.97, (bins < 1e-3).mean()
    rbins = np.abs(rigid.cpu().numpy() - rig) * 10
    assert rbins.max() <= 1.0 + 1e-3 and (rbins < 1e-3).mean() >= 0.9
    # the shifts are not trivial: the video moves by voxels
    assert np.abs(ref[..., :2]).max() > 1.0


def test_known_piecewise_shifts_are_recovered(ops):
    """A video built from known displacements: the recovered patch shifts follow them to within a few tenths of a voxel
    (the registration of a patch that holds blobs from two quadrants is a blend), and frame-to-frame position changes of
    points inside one quadrant match the truth."""
    sz, T = [64, 64, 2], 8
    video, template, pos, truth = synthetic_video(sz, T, 60, seed=5, piecewise=False)
    frames = torch.from_numpy(video.reshape(T, -1)).cuda()
    rigid, patch = ops.register_patches(frames, torch.from_numpy(template).cuda(), sz, (16, 16, 1), (16, 16, 1), (5, 5, 1), 3, 10,
                                        -float(video.min()))
    rigid = rigid.cpu().numpy()
    for t in range(T):
        base = truth[t][0]
        # src = frame, target = template: the shift that registers the template with the frame = the frame's displacement
        np.testing.assert_allclose(rigid[t, :2], base[:2], atol=0.3)
    p = patch.cpu().numpy()
    for t in range(T):   # every patch sees the same rigid displacement; total_shifts = (-x, -y, +z).  (A 32 x 32 x 2 patch holds a
        # handful of blobs, some cut by its border, and its circular correlation is biased by a few tenths of a voxel.)
        for d in range(2):
            assert abs(np.median(p[t, :, d]) + truth[t][0][d]) <= 0.3, (t, d)
            np.testing.assert_allclose(p[t, :, d], -truth[t][0][d], atol=1.0)


def test_motioncorrect_front_end_and_apply_shifts_points(ops, MO):
    """The class surface (reference MotionCorrect.py:64-385): motion_correct_pwrigid fills x/y/z_shifts_els, apply_shifts_points
    returns (K,3,T) positions equal to the oracle's on the same shift tables, and DeformableNMF accepts the first frame's
    positions."""
    from dnmf_amd.Demix.MotionCorrect import MotionCorrect
    from dnmf_amd.Demix import dNMF as M
    sz, T, K = [48, 40, 2], 5, 12
    video, template, pos, _ = synthetic_video(sz, T, 30, seed=9)
    mc = MotionCorrect(video, max_shifts=(5, 5, 1), strides=(16, 12, 1), overlaps=(8, 8, 1), max_deviation_rigid=3, is3D=True,
                       pw_rigid=True)
    mc.motion_correct(template=template)
    assert len(mc.x_shifts_els) == len(mc.y_shifts_els) == len(mc.z_shifts_els) == T
    NP = len(MO.sliding_window_3d(sz, mc.overlaps, mc.strides))
    assert mc.x_shifts_els[0].shape == (NP,) and mc.border_to_0 >= 1
    points = np.random.RandomState(1).rand(K, 3) * np.array(sz)
    P_T = mc.apply_shifts_points(video, points)
    ref = MO.apply_shifts_points(np.stack(mc.x_shifts_els), np.stack(mc.y_shifts_els), np.stack(mc.z_shifts_els), sz, mc.overlaps,
                                 mc.strides, points)
    assert P_T.shape == (K, 3, T)
    np.testing.assert_allclose(P_T, ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(P_T[:, :, 0], points, atol=1e-5)       # frame 0 is the reference frame
    A = mc.apply_shifts_frame(video, points, 2)          # reference :330-349: frame 2's shifts added, all signs +
    idx = ((MO.patch_centers(sz, mc.overlaps, mc.strides)[:, None, :] - points[None]) ** 2).sum(2).argmin(0)
    expect = points + np.stack([mc.x_shifts_els[2][idx], mc.y_shifts_els[2][idx], mc.z_shifts_els[2][idx]], 1)
    np.testing.assert_allclose(A, expect, atol=1e-6)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(P_T[:, :, 0]).float())
    assert tuple(dn.fp.A.shape) == (*sz, K)
    with pytest.raises(NotImplementedError):
        MotionCorrect(video[..., 0], is3D=False).apply_shifts_points(video, points)


@pytest.mark.parametrize("sz,max_shifts", [([48, 40, 2], (5, 5, 1)), ([40, 36, 5], (6, 4, 2)), ([64, 64, 1], (6, 6, 0))])
def test_rigid_correction_vs_oracle(ops, MO, sz, max_shifts):
    """dnmf_rigid_correct against the restatement of tile_and_correct_3d's rigid branch + apply_shifts_dft (:1518-1574,
    :1028-1157): same shifts (to a bin of 0.1 voxel), the frames moved through their spectra equal to 2e-4 of the frame's
    range where both are finite (fp32 transforms against complex128), the same NaN borders -- including the reference's
    pairing of the border of axis 0 with the shift of axis 1 -- and the NaN-aware sums behind the next template."""
    T = 6
    video, template, _, _ = synthetic_video(sz, T, 40, seed=sum(sz) + 1, piecewise=False)
    add = float(np.float32(-video.min()))
    ref_frames, ref_shifts = [], []
    for img in video:
        f, sh = MO.rigid_correct_3d(img, template, max_shifts, 10, add, True)
        ref_frames.append(f), ref_shifts.append(sh)
    ref_frames, ref_shifts = np.array(ref_frames), -np.array(ref_shifts)
    frames = torch.from_numpy(video.reshape(T, -1)).cuda()
    rigid, out, tsum, tcount = ops.rigid_correct(frames, torch.from_numpy(template).cuda(), sz, max_shifts, 10, add, True, want_frames=True)
    rigid, out = rigid.cpu().numpy(), out.cpu().numpy().reshape(T, *sz)
    same = np.abs(rigid - ref_shifts).max(1) < 1e-3
    assert same.mean() >= 0.8 and np.abs(rigid - ref_shifts).max() <= 0.1 + 1e-3
    assert np.abs(ref_shifts[:, :2]).max() > 1.0
    scale = float(video.max() - video.min())
    for t in np.flatnonzero(same):
        np.testing.assert_array_equal(np.isnan(out[t]), np.isnan(ref_frames[t]), err_msg=str(t))
        ok = ~np.isnan(out[t])
        assert ok.any()
        np.testing.assert_allclose(out[t][ok], ref_frames[t][ok], rtol=0, atol=2e-4 * scale, err_msg=str(t))
    np.testing.assert_array_equal(tcount.cpu().numpy().reshape(sz), (~np.isnan(out)).sum(0))
    np.testing.assert_allclose(tsum.cpu().numpy().reshape(sz), np.nansum(out, 0), rtol=1e-5, atol=1e-5)
    # a frame moved by its own shift lines up with the template: the registration of the corrected frame is ~zero
    filled = np.where(np.isnan(out), template[None], out).astype(np.float32)
    r2, _, _, _ = ops.rigid_correct(torch.from_numpy(filled.reshape(T, -1)).cuda(), torch.from_numpy(template).cuda(), sz, max_shifts,
                                    10, add, True)
    assert np.abs(r2.cpu().numpy()).max() <= 0.3


def test_motioncorrect_without_a_template(ops, MO):
    """template=None (reference :298-301): the rigid pre-pass builds the template -- the binned median, one rigid
    correction of every frame against it, the NaN-aware mean of the moved frames -- and the piecewise pass starts from
    there.  Against oracle.rigid_template on the same video; then the class with pw_rigid=False (rigid only)."""
    from dnmf_amd.Demix.MotionCorrect import MotionCorrect
    sz, T = [48, 40, 2], 23            # 23 frames: two bins of ten, three frames left out of the first template
    video, _, _, _ = synthetic_video(sz, T, 30, seed=21, piecewise=False)
    templ, shifts, moved = MO.rigid_template(video, (5, 5, 1), add_to_movie=-float(video.min()))
    mc = MotionCorrect(video, max_shifts=(5, 5, 1), strides=(16, 12, 1), overlaps=(8, 8, 1), is3D=True, pw_rigid=False,
                       save_corrected=True)
    mc.motion_correct()
    got = np.array(mc.shifts_rig)
    assert got.shape == (T, 3)
    bins = np.abs(got - shifts) * 10
    assert bins.max() <= 1.0 + 1e-3 and (bins < 1e-3).mean() >= 0.85
    scale = float(video.max() - video.min())
    # (a frame whose shift landed in the neighbouring bin moves the mean by 0.1 voxel / T of a gradient)
    np.testing.assert_allclose(mc.total_template_rig, templ, rtol=0, atol=5e-3 * scale)
    assert np.abs(mc.total_template_rig - templ).mean() <= 2e-4 * scale
    assert mc.mc[0].shape == (*sz, T) and mc.border_to_0 >= 1
    same = np.flatnonzero(bins.max(1) < 1e-3)
    t = int(same[0])
    ok = ~np.isnan(moved[t])
    np.testing.assert_allclose(mc.mc[0][..., t][ok], moved[t][ok], atol=2e-4 * scale)
    first = mc._bin_median_3d(torch.from_numpy(video.reshape(T, -1)).cuda()).cpu().numpy().reshape(sz)
    np.testing.assert_allclose(first, MO.bin_median_3d(video), rtol=1e-6, atol=1e-7)
    # the piecewise pass on top of it
    mc2 = MotionCorrect(video, max_shifts=(5, 5, 1), strides=(16, 12, 1), overlaps=(8, 8, 1), is3D=True, pw_rigid=True)
    mc2.motion_correct()
    np.testing.assert_allclose(mc2.total_template_els.cpu().numpy(), mc.total_template_rig, atol=1e-6)
    sx, sy, sz_, _ = MO.pw_rigid_shifts(video, mc.total_template_rig, (16, 12, 1), (8, 8, 1), (5, 5, 1), 10, 3, -float(video.min()))
    pb = np.abs(np.stack(mc2.x_shifts_els) - sx) * 10
    assert pb.max() <= 1.0 + 1e-3 and (pb < 1e-3).mean() >= 0.95


@pytest.mark.parametrize("border", [False, "min", "copy"])
def test_rigid_correction_border_modes(ops, MO, border):
    """border_nan False / 'min' / 'copy' of apply_shifts_dft (:1098-1145) against the restatement."""
    sz, max_shifts, T = [40, 36, 5], (6, 4, 2), 5
    video, template, _, _ = synthetic_video(sz, T, 40, seed=77, piecewise=False)
    add = float(np.float32(-video.min()))
    ref = [MO.rigid_correct_3d(img, template, max_shifts, 10, add, border) for img in video]
    frames = torch.from_numpy(video.reshape(T, -1)).cuda()
    rigid, out, tsum, tcount = ops.rigid_correct(frames, torch.from_numpy(template).cuda(), sz, max_shifts, 10, add, border, want_frames=True)
    rigid, out = rigid.cpu().numpy(), out.cpu().numpy().reshape(T, *sz)
    scale = float(video.max() - video.min())
    checked = 0
    for t in range(T):
        if np.abs(rigid[t] + np.array(ref[t][1])).max() > 1e-3:      # (ref holds the flipped sign)
            continue
        assert not np.isnan(out[t]).any()
        np.testing.assert_allclose(out[t], ref[t][0], rtol=0, atol=3e-4 * scale, err_msg=str(t))
        checked += 1
    assert checked >= 3
    assert int(tcount.min()) == T


def test_two_dimensional_video(ops, MO):
    """is3D=False: a (T, X, Y) video through register_translation / tile_and_correct (reference :801-1024, :1272-1418) =
    the same kernels on one slice; x_shifts_els / y_shifts_els against oracle.tile_shifts_2d."""
    from dnmf_amd.Demix.MotionCorrect import MotionCorrect
    sz, T = [64, 56, 1], 5
    video, template, _, _ = synthetic_video(sz, T, 40, seed=31)
    video2, template2 = video[..., 0], template[..., 0]
    mc = MotionCorrect(video2, max_shifts=(6, 6), strides=(24, 20), overlaps=(8, 8), max_deviation_rigid=3, is3D=False, pw_rigid=True)
    mc.motion_correct(template=template2)
    assert not hasattr(mc, "z_shifts_els") and len(mc.x_shifts_els) == T and len(mc.coord_shifts_els[0][0]) == 2
    ref = np.array([MO.tile_shifts_2d(img, template2, (24, 20), (8, 8), (6, 6), 10, 3, -float(video2.min()))[1] for img in video2])
    got = np.stack([np.stack(mc.x_shifts_els), np.stack(mc.y_shifts_els)], 2)
    bins = np.abs(got - ref) * 10
    assert bins.max() <= 1.0 + 1e-3 and (bins < 1e-3).mean() >= 0.95
    assert np.abs(ref).max() > 1.0 and tuple(mc.total_template_els.shape) == (64, 56)
    with pytest.raises(NotImplementedError):
        mc.motion_correct_rigid()


def test_matrix_pipe_kernel_equals_the_vector_kernel(ops, monkeypatch):
    """K8's axis transform runs as f32 MFMAs (mc_axis_mfma_kernel); DNMF_K8_VALU=1 selects the vector-ALU kernel it replaced.
    Same shifts (a bin apart at most, 98 % identical: the sums over j run in a different order) and the same moved frames to
    1e-5 of the range, on a volume with odd sizes, a patch grid and every pass (forward, windowed inverse, upsampled inverse,
    the full inverse on a shifted grid)."""
    sz, T = [72, 50, 3], 6
    video, template, _, _ = synthetic_video(sz, T, 40, seed=12)
    frames, tm = torch.from_numpy(video.reshape(T, -1)).cuda(), torch.from_numpy(template).cuda()
    add = -float(video.min())
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("DNMF_K8_VALU", mode)
        rigid, patch = ops.register_patches(frames, tm, sz, (20, 14, 2), (10, 8, 1), (6, 5, 1), 3, 10, add)
        r2, out, _, _ = ops.rigid_correct(frames, tm, sz, (6, 5, 1), 10, add, False, want_frames=True)
        res[mode] = (rigid.cpu().numpy(), patch.cpu().numpy(), r2.cpu().numpy(), out.cpu().numpy())
    for a, b in zip(res["0"][:3], res["1"][:3]):
        assert np.abs(a - b).max() <= 0.1 + 1e-4 and (np.abs(a - b) < 1e-4).mean() >= 0.98
    same = np.abs(res["0"][2] - res["1"][2]).max(1) < 1e-4
    scale = float(video.max() - video.min())
    np.testing.assert_allclose(res["0"][3][same], res["1"][3][same], rtol=0, atol=1e-5 * scale)
