"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (fp32 path vs the reference's fp32 warp / float64 NMF arithmetic):
  A_t, A_tC   atol 2e-6   interpolation weights differ in rounding order only (values are O(1))
  beta.grad   rtol 1e-4 of the largest component (sums over P voxels in a different order)
  G, r        rtol 2e-5   fp32 MFMA chains + ordered chunk sums vs float64 einsum
  C           rtol 1e-4   after up to 50 multiplicative rounds
"""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from dnmf_amd.Demix import dNMF
    return dNMF


@pytest.fixture(scope="module")
def O():
    from oracle import dnmf_oracle
    return dnmf_oracle


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def border_is_zero(img, sz):
    """True when every float outside the volume of (rows of) halo-layout images is an exact zero."""
    from dnmf_amd import ops
    c = img[..., :ops.halo_voxels(sz)].clone()
    ops.halo_interior(c, sz).zero_()
    return int(torch.count_nonzero(c)) == 0


def make_fp(M, sz, K, T, positions, beta=None, A=None):
    fp = M.ExponentialFP(torch.as_tensor(np.asarray(sz)), K, T, positions=torch.as_tensor(np.asarray(positions)).float())
    if A is not None:
        fp.A = dev(A)
    if beta is not None:
        with torch.no_grad():
            fp.beta.copy_(dev(beta))
    return fp


def test_G1_constructor(M):
    g = golden("G1_init")
    fp = make_fp(M, g["sz"], 3, 5, g["positions"])
    np.testing.assert_array_equal(fp.flow_id.cpu().numpy(), g["lattice"])
    np.testing.assert_array_equal(fp.transformed.cpu().numpy(), g["basis"])
    np.testing.assert_array_equal(fp.beta.detach().cpu().numpy(), g["beta"])
    assert fp.beta.requires_grad and fp.beta.is_leaf
    np.testing.assert_allclose(fp.A.cpu().numpy(), g["A"], rtol=2e-6, atol=1e-30)  # device exp vs host exp


def test_G2_forward(M, O):
    g = golden("G2_forward")
    A = O.gaussian_footprints(g["sz"], g["positions"], np.full(3, 3.0))
    fp = make_fp(M, g["sz"], 3, 5, g["positions"], beta=g["beta"], A=A)
    A_tC, A_t, grid, reg = fp(g["times"].tolist(), torch.from_numpy(g["C"]))
    assert tuple(A_t.shape) == g["A_t"].shape and tuple(grid.shape) == g["grid"].shape
    np.testing.assert_allclose(grid.cpu().numpy(), g["grid"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(A_t.cpu().numpy(), g["A_t"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(A_tC.detach().cpu().numpy(), g["A_tC"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(reg.cpu().numpy(), g["reg"], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("label", ["id_b1", "id_b3", "pert_b1", "pert_b3", "pert_b4"])
def test_G3_beta_grad(M, O, label):
    """Both routes to beta.grad: autograd through forward() and the fused mini-batch kernel."""
    g = golden("G3_grad")
    A = O.gaussian_footprints(g["sz"], g["positions"], np.full(4, 3.0))
    times = g[label + "_times"].tolist()
    frames = dev(np.moveaxis(g["video"][..., times], -1, 0))
    want = g[label + "_grad"]
    tol = 1e-4 * np.abs(want).max()

    fp = make_fp(M, g["sz"], 4, 8, g["positions"], beta=g[label + "_beta"], A=A)
    A_tC, _, _, _ = fp(times, torch.from_numpy(g["C"]))
    loss = torch.nn.functional.mse_loss(A_tC, frames)
    loss.backward()
    np.testing.assert_allclose(float(loss), g[label + "_loss"], rtol=1e-5)
    np.testing.assert_allclose(fp.beta.grad.cpu().numpy(), want, rtol=1e-4, atol=tol)

    from dnmf_amd import ops
    fp2 = make_fp(M, g["sz"], 4, 8, g["positions"], beta=g[label + "_beta"], A=A)
    S = ops.recon_image(fp2.packed_footprints(), 4, fp2.sz_list, dev(g["C"]), times)
    grad = torch.zeros_like(fp2.beta)
    out = ops.warp_recon_grad(S, None, frames.reshape(len(times), -1), None, fp2.sz_list, fp2.beta.detach(), times,
                              grad=grad)
    np.testing.assert_allclose(float(out["loss"][0]), g[label + "_loss"], rtol=1e-5)
    np.testing.assert_allclose(grad.cpu().numpy(), want, rtol=1e-4, atol=tol)


def test_G6_gram_rhs_and_G4_temporal(M, O):
    from dnmf_amd import ops
    g6, g4 = golden("G6_pushforward"), golden("G4_temporal")
    A = O.gaussian_footprints(g6["sz"], g6["positions"], np.full(4, 3.0))
    fp = make_fp(M, g6["sz"], 4, 8, g6["positions"], beta=g6["beta"], A=A)
    frames = dev(np.moveaxis(g6["video"], -1, 0)).reshape(8, -1)
    G, r, _ = ops.warp_gram_rhs(fp.packed_footprints(), 4, fp.sz_list, fp.beta.detach(), list(range(8)), frames)
    Gref, rref = O.gram_rhs(g6["A_t"].astype(np.float64), g6["Y"].astype(np.float64))
    np.testing.assert_allclose(G.cpu().numpy(), np.moveaxis(Gref, 2, 0), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r.cpu().numpy(), rref.T, rtol=2e-5, atol=1e-7)
    for label, gamma in (("none", None), ("zero", 0), ("g1e2", 1e-2)):
        C1 = M._mu_temporal(G, r, dev(g4["C0"], torch.float64), gamma, 1)
        np.testing.assert_allclose(C1.cpu().numpy(), g4[label + "_it1"], rtol=5e-5)
        C50 = M._mu_temporal(G, r, dev(g4["C0"], torch.float64), gamma, 50)
        np.testing.assert_allclose(C50.cpu().numpy(), g4[label + "_it50"], rtol=1e-4)
    C50f = M._mu_temporal(G, r, dev(g4["C0"]), 0, 50)          # fp32 state, fused loop (update_footprints' route)
    np.testing.assert_allclose(C50f.cpu().numpy(), g4["zero_it50"], rtol=1e-4)
    # static update_temporal on an explicit A_t, numpy in / numpy out
    C1 = M.DeformableNMF.update_temporal(g6["A_t"].astype(np.float64), g4["C0"], g6["Y"].astype(np.float64), gamma=1e-2)
    np.testing.assert_allclose(C1, g4["g1e2_it1"], rtol=5e-5)


def test_G5_update_spatial(M):
    g = golden("G5_spatial")
    # K5 (fp32 MFMA over the frames) + K6 against the reference's float64 einsums: rtol 2e-5
    np.testing.assert_allclose(M.DeformableNMF.update_spatial(g["A"], g["C"], g["Y_i"]), g["out_noD"], rtol=2e-5)
    np.testing.assert_allclose(M.DeformableNMF.update_spatial(g["A"], g["C"], g["Y_i"], D=g["D"], gamma=0.7), g["out_D"],
                               rtol=2e-5)


@pytest.mark.timeout(300)
def test_rccl_communicator_single_rank(M):
    """C1 through the C ABI on the one GPU of this box: a one-rank RCCL communicator whose all-reduce leaves the
    buffer as it was, on torch's stream, between two kernels of the path.  (More than one rank needs one GPU per
    rank; the two-rank exchange is covered with gloo in test_gpu_multiprocess.py.)"""
    from dnmf_amd import ops
    comm = ops.Communicator(None)
    assert (comm.nranks, comm.rank) == (1, 0)
    torch.manual_seed(0)
    Y, C = torch.rand(16, 1000, device="cuda"), torch.rand(20, 16, device="cuda")
    A1, Cs = ops.spatial_accum(Y, C)
    ref1, ref2 = A1.clone(), Cs.clone()
    comm.all_reduce_(A1)
    comm.all_reduce_(Cs)
    torch.cuda.synchronize()
    assert torch.equal(A1, ref1) and torch.equal(Cs, ref2)
    with pytest.raises(ValueError):
        comm.all_reduce_(A1.double())
    comm.close()
    comm.close()  # idempotent


def test_spatial_accum_chunks_and_shards(M):
    """A1 / C_s accumulated over two frame chunks (what two T-shards all-reduce) equal the one-shot result, and
    both equal float64 matmuls; K up to 128 and a ragged voxel count."""
    from dnmf_amd import ops
    torch.manual_seed(0)
    for P, K, T in ((1000, 100, 37), (4096, 20, 64), (77, 128, 9)):
        Y = torch.rand(T, P, device="cuda")
        C = torch.rand(K, T, device="cuda")
        A1, Cs = ops.spatial_accum(Y, C)
        want1 = (Y.double().T @ C.double().T)
        wants = C.double() @ C.double().T
        assert float((A1.double() - want1).abs().max()) < 2e-5 * float(want1.abs().max())
        assert float((Cs.double() - wants).abs().max()) < 2e-6 * float(wants.abs().max())
        h = T // 2
        a, c = ops.spatial_accum(Y, C, frame_ids=list(range(h)), times=list(range(h)))
        ops.spatial_accum(Y, C, frame_ids=list(range(h, T)), times=list(range(h, T)), A1=a, Cs=c, accumulate=True)
        assert float((a - A1).abs().max()) < 1e-5 * float(A1.abs().max())
        A = torch.rand(P, K, device="cuda")
        D = torch.rand(P, K, device="cuda")
        want = A.double() * A1.double() / (A.double() @ Cs.double() + 0.3 * D.double() + 1e-32)
        got = ops.mu_spatial(A.clone(), A1, Cs, D, 0.3)
        assert float(((got.double() - want) / want).abs().max()) < 2e-5


def test_G6_pushforward_surface(M, O):
    g = golden("G6_pushforward")
    A = O.gaussian_footprints(g["sz"], g["positions"], np.full(4, 3.0))
    dn = M.DeformableNMF(torch.from_numpy(g["sz"]), 4, 8, positions=torch.from_numpy(g["positions"]))
    dn.fp.A = dev(A)
    with torch.no_grad():
        dn.fp.beta.copy_(dev(g["beta"]))
    np.testing.assert_allclose(dn.D, g["D"], rtol=1e-10)
    frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(g["video"], -1, 0)))
    loader = [(frames[s:s + 3], torch.arange(s, min(8, s + 3))) for s in range(0, 8, 3)]
    A_t, Y_i, Y = M.ExponentialFP.spatial_pushforward(loader, 3, g["sz"], "cuda", dn)
    np.testing.assert_allclose(A_t, g["A_t"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(Y.astype(np.float32), g["Y"])
    # registered video (K7): identical to scipy's nearest-neighbour result except where two warped voxels are
    # (nearly) equidistant from a lattice point -- cKDTree's choice between them is unspecified
    # (at the identity every z = 1 lattice point is exactly midway between warped slices 0 and 2: sz/(sz-1) = 2)
    mism = Y_i.astype(np.float32) != g["Y_i"]
    assert not mism[..., [1, 2, 6, 7]].any() or mism.mean() < 0.2
    lat = O.voxel_lattice(g["sz"]).reshape(-1, 3).astype(np.float64)
    basis = O.quadratic_basis(O.voxel_lattice(g["sz"]))
    for t in range(8):
        if not mism[..., t].any():
            continue
        _, n = O.poly_grid(basis, g["beta"][:, :, [t]], g["sz"])
        pts = O.pushforward_flow(n, g["sz"])[..., 0].reshape(-1, 3).astype(np.float64)
        for q in np.flatnonzero(mism[..., t].reshape(-1)):
            d = np.sort(((pts - lat[q]) ** 2).sum(1))
            assert d[1] - d[0] < 1e-6, (t, q, d[:3])


@pytest.mark.parametrize("sz", [[12, 10, 2], [64, 48, 1], [40, 36, 3], [160, 128, 1], [96, 64, 2], [48, 40, 5]])
def test_registered_video_window_search_equals_exhaustive(M, O, sz):
    """K7's window search must return exactly what the exhaustive search (every voxel a candidate for every lattice
    point, the kernel of round 1) returns: identity (every lattice point of an odd slice is a tie at Z = 2), the
    warps of fixture G6, mild random warps, shifts that leave a band of lattice points far from every warped voxel, and
    warps strong enough to fold (those take the exhaustive fall-back, counted)."""
    from dnmf_amd import ops
    rng = np.random.RandomState(sum(sz))
    T = 14
    beta = O.identity_beta(T)
    amp = np.array([0.0, 0.0] + list(np.logspace(-2, 1.3, T - 2)))
    base = np.array([3.0, 2e-2, 2e-2, 2e-2, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4])
    beta += (rng.randn(10, 3, T) * base[:, None, None] * amp[None, None, :]).astype(np.float32)
    beta[0, 0, 1] = 7.3                       # frame 1: a pure shift of 7.3 voxels along x
    if sz[2] == 1:
        beta[:, 2] = O.identity_beta(T)[:, 2]
    if sz == [12, 10, 2]:
        g6 = golden("G6_pushforward")["beta"]
        beta[:, :, 2:2 + min(8, T - 2)] = g6[:, :, :min(8, T - 2)]
    frames = torch.rand(T, int(np.prod(sz)), device="cuda")
    b = dev(beta)
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    win = ops.image_iwarp(frames, None, sz, b, list(range(T)), count=count)
    full = ops.image_iwarp(frames, None, sz, b, list(range(T)), exhaustive=True)
    assert torch.equal(win, full)
    # mild warps are served by the window search alone
    count.zero_()
    ops.image_iwarp(frames, None, sz, b, [0, 2, 3, 4], count=count)
    assert int(count) == 0, int(count)
    # row ids, an output with a wider row stride
    out = torch.full((3, int(np.prod(sz)) + 5), -1.0, device="cuda")
    ops.image_iwarp(frames, [5, 0, 9], sz, b, [4, 6, 8], out=out)
    ref = ops.image_iwarp(frames[[5, 0, 9]].contiguous(), None, sz, b, [4, 6, 8], exhaustive=True)
    assert torch.equal(out[:, :-5], ref) and bool((out[:, -5:] == -1.0).all())


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("label", ["lr1e-5_ordered", "lr1e-3_ordered", "lr1e-3_shuffled"])
def test_G9_demo_loop(M, O, label, fused):
    """Fixture G9 with host loaders (lists of (frames, indices), as a DataLoader yields them): ``fused`` = the epoch
    staged on the GPU and evaluated by the per-column Adam kernel, else one K2 launch and one ``optimizer.step()`` of
    torch per mini-batch."""
    g = golden("G9_loop")
    lr, epochs, shuffled, bs = g[label + "_cfg"]
    bs = int(bs)
    A = O.gaussian_footprints(g["sz"], g["positions"], np.full(4, 3.0))
    dn = M.DeformableNMF(torch.from_numpy(g["sz"]), 4, 8, positions=torch.from_numpy(g["positions"]))
    dn.verbose = False
    dn.fused_motion = fused
    dn.fp.A = dev(A)
    dn.C = dev(g[label + "_C0"])
    opt = torch.optim.Adam([dn.fp.beta], lr=float(lr))
    frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(g["video"], -1, 0)))
    order = [[[i for i in b if i >= 0] for b in ep] for ep in g[label + "_order"].tolist()]
    for ep in order:
        dn.update_motion([(frames[b], torch.tensor(b)) for b in ep], opt, gamma=1, epochs=1)
    ident = O.identity_beta(8)
    want = g[label + "_beta_after_motion"]
    scale = np.abs(want - ident).max()
    # Adam moves every coefficient by ~lr per step whatever the gradient's size: compare the displacement
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - ident, want - ident, rtol=0, atol=2e-3 * scale)
    test = [(frames[s:s + bs], torch.arange(s, s + bs)) for s in range(0, 8, bs)]
    A_t, Y_i, Y = dn.update_footprints(test, bs, g["sz"], gamma_c=0, iter_c=5)
    assert A_t.shape == (20, 16, 2, 4, 8) and Y.shape == (20, 16, 2, 8) and Y_i.shape == Y.shape
    np.testing.assert_allclose(dn.C.cpu().numpy(), g[label + "_C_after_footprints"], rtol=1e-4)


def test_G10_z1_semantics(M):
    g = golden("G10_2d")
    X, Y, _ = g["sz"]
    T = g["beta"].shape[2]
    fp = make_fp(M, [int(X), int(Y), 1], 3, T, np.zeros((3, 3), np.float32), beta=g["beta"], A=g["A2d"][:, :, None, :])
    times = list(range(T))
    A_tC, A_t, grid, _ = fp(times, torch.from_numpy(g["C"]))
    np.testing.assert_allclose(A_t.cpu().numpy()[..., 0], g["A_t"][..., 0], rtol=0, atol=2e-6)
    np.testing.assert_allclose(A_tC.detach().cpu().numpy()[..., 0], g["A_tC"][..., 1], rtol=0, atol=2e-6)
    frames = dev(np.moveaxis(g["Y2d"], -1, 0))[:, :, :, None]
    loss = torch.nn.functional.mse_loss(A_tC, frames)
    loss.backward()
    np.testing.assert_allclose(float(loss), g["loss"], rtol=1e-5)
    grad = fp.beta.grad.cpu().numpy()
    noz = [0, 1, 2, 4, 5, 7]
    np.testing.assert_allclose(grad[noz, :2], g["grad"][noz, :2], rtol=1e-4, atol=1e-4 * np.abs(g["grad"]).max())
    assert np.all(grad[:, 2] == 0) and np.all(grad[[3, 6, 8, 9]] == 0)
    from dnmf_amd import ops
    G, r, _ = ops.warp_gram_rhs(fp.packed_footprints(), 3, fp.sz_list, fp.beta.detach(), times, frames.reshape(T, -1))
    C5 = M._mu_temporal(G, r, dev(g["C"]), 0, 5)
    np.testing.assert_allclose(C5.cpu().numpy(), g["C_it5"], rtol=1e-4)


@pytest.mark.parametrize("sz,K,T", [([40, 36, 3], 20, 5), ([33, 47, 1], 50, 6), ([48, 40, 2], 100, 4),
                                    ([21, 19, 2], 10, 9), ([32, 32, 1], 120, 3), ([5, 66000, 1], 3, 2)])
def test_random_problem_vs_oracle(M, O, sz, K, T):
    """Every padded-K variant of the Gram kernel (NB = 1..8), ragged voxel counts, 2-D and 3-D, an axis longer than
    65536 (IEEE division instead of the shortcut in K2)."""
    from dnmf_amd import ops
    rng = np.random.RandomState(K)
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, 3.0))
    beta = O.identity_beta(T)
    sc = min(1.0, 64.0 / max(sz))   # keep the displacement of the far end of a long axis at a few voxels
    beta += (rng.randn(10, 3, T) * np.array([0.7] + [1e-2 * sc] * 3 + [2e-4 * sc * sc] * 6)[:, None, None]).astype(np.float32)
    if sz[2] == 1:
        beta[:, 2] = O.identity_beta(T)[:, 2]
    C = rng.rand(K, T).astype(np.float32)
    video = rng.rand(*sz, T).astype(np.float32)
    lat = O.voxel_lattice(sz)
    basis = O.quadratic_basis(lat)
    times = list(range(T))
    A_tC, A_t, n, reg = O.forward(A, basis, beta, sz, times, C, O.trilinear_sample_torch)
    fp = make_fp(M, sz, K, T, pos, beta=beta, A=A)
    gA_tC, gA_t, ggrid, greg = fp(times, torch.from_numpy(C))
    # A coordinate near 66000 has an fp32 spacing of 0.008 voxels: there the order in which the ten polynomial terms are
    # added (torch's einsum vs the fused chain) moves the interpolation weights by that much, so the tolerances scale
    # with the spacing at the far end of the longest axis (factor 1 for the ordinary volumes)
    w = max(1.0, 0.5 * float(np.spacing(np.float32(max(sz)))) / 5e-6)
    # 1e-5: the tolerance SURVEY 8(c) states for A_t (either evaluation is within 5e-6 of exact arithmetic)
    np.testing.assert_allclose(gA_t.cpu().numpy(), A_t, rtol=0, atol=1e-5 * w)
    np.testing.assert_allclose(gA_tC.detach().cpu().numpy(), A_tC, rtol=1e-5, atol=2e-5 * w)
    np.testing.assert_allclose(greg.cpu().numpy(), reg, rtol=1e-3, atol=1e-7)
    frames = np.moveaxis(video, -1, 0)
    loss, grad = O.mse_beta_grad_autograd(A, basis, beta, sz, times, C, frames)
    gl = torch.nn.functional.mse_loss(gA_tC, dev(frames))
    gl.backward()
    np.testing.assert_allclose(float(gl), loss, rtol=1e-5 * w)
    np.testing.assert_allclose(fp.beta.grad.cpu().numpy(), grad, rtol=1e-4 * w, atol=1e-4 * w * np.abs(grad).max())
    G, r, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), times, dev(frames).reshape(T, -1))
    A64 = np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0])
    Gref, rref = O.gram_rhs(A64, video.astype(np.float64))
    np.testing.assert_allclose(G.cpu().numpy(), np.moveaxis(Gref, 2, 0), rtol=2e-5 * w, atol=2e-5 * w * np.abs(Gref).max())
    np.testing.assert_allclose(r.cpu().numpy(), rref.T, rtol=2e-5 * w, atol=2e-5 * w * np.abs(rref).max())
    Cref = O.mu_temporal_from_gram(Gref, rref, C, None, 20)
    Cg = M._mu_temporal(G, r, dev(C), None, 20)
    np.testing.assert_allclose(Cg.cpu().numpy(), Cref, rtol=2e-4 * w, atol=1e-7)


@pytest.mark.parametrize("sz,K,T", [([40, 36, 3], 20, 5), ([33, 47, 1], 50, 6), ([48, 40, 2], 100, 4),
                                    ([21, 19, 2], 10, 3), ([32, 32, 1], 120, 3), ([70, 64, 1], 100, 2)])
def test_bf16_gram_vs_rounded_oracle(M, O, sz, K, T):
    """K3b (BASELINE config 5's "bf16 MFMA"; SURVEY 8(c): Gram inputs rounded to bf16, fp32 accumulate).  The oracle
    is the reference contraction on the oracle's A_t and the frame, both rounded to bf16 (nearest even): the kernel
    must match that up to the A_t elements that sit on a rounding boundary and round the other way (the blend differs
    from the oracle's by up to 5e-6), each of which moves its products by 2^-8 of themselves: 1e-3 of the largest
    entry (observed 4e-4 on the 32x32 volume, where an entry is a sum over few voxels).  Against the unrounded
    contraction the stated tolerance is 1e-2 (of the largest entry) on G, r and on C after 20 updates."""
    from dnmf_amd import ops
    rng = np.random.RandomState(K + 1)
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, 3.0))
    beta = O.identity_beta(T)
    beta += (rng.randn(10, 3, T) * np.array([0.7, 1e-2, 1e-2, 1e-2, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4])[:, None, None]
             ).astype(np.float32)
    if sz[2] == 1:
        beta[:, 2] = O.identity_beta(T)[:, 2]
    C = rng.rand(K, T).astype(np.float32)
    video = rng.rand(*sz, T).astype(np.float32)
    times = list(range(T))
    _, A_t, _, _ = O.forward(A, O.quadratic_basis(O.voxel_lattice(sz)), beta, sz, times, C, O.trilinear_sample_torch)

    def rounded(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().double().numpy()

    A64 = np.transpose(A_t, [2, 3, 4, 1, 0])
    Gb, rb = O.gram_rhs(rounded(A64), rounded(video))
    Gx, rx = O.gram_rhs(A64.astype(np.float64), video.astype(np.float64))
    fp = make_fp(M, sz, K, T, pos, beta=beta, A=A)
    frames = dev(np.moveaxis(video, -1, 0)).reshape(T, -1)
    G, r, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), times, frames, bf16=True)
    G, r = G.cpu().numpy(), r.cpu().numpy()
    np.testing.assert_allclose(G, np.moveaxis(Gb, 2, 0), rtol=0, atol=1e-3 * np.abs(Gb).max())
    np.testing.assert_allclose(r, rb.T, rtol=0, atol=1e-3 * np.abs(rb).max())
    np.testing.assert_allclose(G, np.moveaxis(Gx, 2, 0), rtol=0, atol=1e-2 * np.abs(Gx).max())
    np.testing.assert_allclose(r, rx.T, rtol=0, atol=1e-2 * np.abs(rx).max())
    assert np.array_equal(G, np.transpose(G, (0, 2, 1)))
    # through the model switch, and the traces it leads to
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
    dn.fp.A = dev(A)
    with torch.no_grad():
        dn.fp.beta.copy_(dev(beta))
    dn.gram_kernel = 'bf16'
    G2, r2 = dn._gram_rhs(frames, torch.arange(T, dtype=torch.int32, device="cuda"))
    assert np.array_equal(G2.cpu().numpy(), G) and np.array_equal(r2.cpu().numpy(), r)
    Cb = M._mu_temporal(G2, r2, dev(C), None, 20).cpu().numpy()
    Cx = O.mu_temporal_from_gram(Gx, rx, C, None, 20)
    np.testing.assert_allclose(Cb, Cx, rtol=0, atol=1e-2 * np.abs(Cx).max())


def test_full_size_properties(M):
    """512x512, K=100 (BASELINE config 3 geometry) through properties that need no CPU reference:
    identity warp => the Gram matrix of every frame is A^T A; integer translation => Gram of the shifted
    footprints; r is linear in the frame."""
    from dnmf_amd import ops
    torch.manual_seed(0)
    sz, K, T = [512, 512, 1], 100, 6
    pos = torch.rand(K, 3) * torch.tensor([512.0, 512.0, 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    frames = torch.rand(T, 512 * 512, device="cuda")
    with torch.no_grad():
        fp.beta[0, 0, 1] = 3.0       # frame 1: shift +3 px in x
        fp.beta[0, 1, 2] = -2.0      # frame 2: shift -2 px in y
    G, r, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), list(range(T)), frames)
    A2 = fp.A.reshape(-1, K).double()
    G0 = (A2.T @ A2)
    scale = float(G0.abs().max())
    for t in (0, 3, 5):
        assert float((G[t].double() - G0).abs().max()) < 2e-5 * scale
        rt = A2.T @ frames[t].double()
        assert float((r[t].double() - rt).abs().max()) < 2e-5 * float(rt.abs().max())
    A3 = fp.A.reshape(512, 512, K)
    sh = torch.zeros_like(A3)
    sh[:-3] = A3[3:]                 # A_t(x,y) = A(x+3, y), zero beyond the edge
    S2 = sh.reshape(-1, K).double()
    assert float((G[1].double() - S2.T @ S2).abs().max()) < 2e-5 * scale
    sh = torch.zeros_like(A3)
    sh[:, 2:] = A3[:, :-2]
    S2 = sh.reshape(-1, K).double()
    assert float((G[2].double() - S2.T @ S2).abs().max()) < 2e-5 * scale
    # linearity of the rhs in the frame
    mix = (2.0 * frames[3] - 0.5 * frames[4])[None].contiguous()
    with torch.no_grad():
        b1 = fp.beta.detach()[:, :, 3:4].contiguous()
    _, rmix, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, b1, [0], mix)
    want = 2.0 * r[3] - 0.5 * r[4]
    assert float((rmix[0] - want).abs().max()) < 1e-4 * float(want.abs().max())
    # G symmetric and the multiplicative update keeps traces non-negative
    assert torch.equal(G, G.transpose(1, 2))
    C = M._mu_temporal(G, r, torch.rand(K, T, device="cuda"), None, 50)
    assert bool((C >= 0).all()) and bool(torch.isfinite(C).all())


def test_full_size_motion_epoch_fused_vs_stepwise(M):
    """update_motion as the bench runs it (512x512, K=100, shuffled mini-batches of 4, torch Adam at the bench's step
    size; 600 frames = 150 optimiser steps per epoch, beyond the 64 that the fused epoch takes literally, so the
    closed-form coasting is in play) against the step-by-step evaluation -- one K2 launch and one optimizer.step() per
    mini-batch: the displacement of beta agrees to 2e-3 of the largest displacement, the stated Adam tolerance."""
    from dnmf_amd.WUtils import Simulator
    torch.manual_seed(5)
    np.random.seed(5)
    sz, K, T = [512, 512, 1], 100, 600
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    out = []
    for fused in (True, False):
        torch.manual_seed(6)
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=positions[:, :, 0].contiguous())
        dn.verbose, dn.fused_motion = False, fused
        beta0 = dn.fp.beta.detach().clone()
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-5 * (50.0 / 512) ** 2)
        loader = M.ResidentLoader(frames, sz, 4, shuffle=True, generator=torch.Generator().manual_seed(7))
        dn.update_motion(loader, opt, gamma=1, epochs=2)
        out.append(dn.fp.beta.detach() - beta0)
    da, db = out
    assert torch.isfinite(da).all() and float(db.abs().max()) > 0
    # the three unit coefficients of the identity are left out: at this step size (9.5e-8) an Adam increment is below
    # their fp32 spacing (1.2e-7), so a run that steps 300 times rounds each of them 300 times and the closed form does
    # not (test_adam_epoch_against_torch_steps measures both against float64 stepping)
    free = torch.ones_like(da, dtype=torch.bool)
    free[1, 0], free[2, 1], free[3, 2] = False, False, False
    diff = float((da - db)[free].abs().max())
    assert diff < 2e-3 * float(db[free].abs().max()), diff / float(db[free].abs().max())
    assert float((da - db).abs().max()) < 16 * 1.2e-7


def test_full_size_long_video_lists_vs_dense(M):
    """The Gram step as the bench runs it -- 512x512, K=100, enough frames (1700) for the library to choose the two
    launches with ~100 tiles per wave by itself -- against the dense MFMA kernel on the same warped problem: G, r entry by
    entry on a spread of frames, and the traces after update_footprints through either kernel."""
    from dnmf_amd import ops
    from dnmf_amd.WUtils import Simulator
    torch.manual_seed(3)
    np.random.seed(3)
    sz, K, T = [512, 512, 1], 100, 1700
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    assert ops._lib.load().dnmf_warp_gram_rhs_lists_chunks(512, 512, 1, T) == 2 * 10   # two launches, ten chunks each

    def model(kernel):
        torch.manual_seed(4)
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=positions[:, :, 0].contiguous())
        dn.verbose, dn.gram_kernel = False, kernel
        with torch.no_grad():   # a smooth warp per frame, a few voxels at the far corner, off the lattice
            t = torch.arange(T, device=dn.fp.beta.device, dtype=torch.float32)
            dn.fp.beta[0, 0] += 0.37 + 1.5 * torch.sin(t / 90.0)
            dn.fp.beta[0, 1] -= 0.21 + 1.1 * torch.cos(t / 70.0)
            dn.fp.beta[1, 0] += 2e-3 * torch.sin(t / 50.0)
            dn.fp.beta[7, 1] += 6e-6 * torch.cos(t / 40.0)
        return dn

    a, b = model('lists'), model('dense')
    pick = torch.tensor([0, 1, 2, 411, 850, 1288, 1699], dtype=torch.int32, device="cuda")
    Gn, rn, _ = ops.warp_gram_rhs_lists(a.fp.packed_lists(), K, sz, a.fp.beta.detach(), None, frames)
    Gd, rd = ops.warp_gram_rhs(b.fp.packed_footprints(), K, sz, b.fp.beta.detach(), pick, frames, frame_ids=pick)[:2]
    scale = float(Gd.abs().max())
    assert float((Gn[pick.long()] - Gd).abs().max()) < 2e-6 * scale
    assert float((rn[pick.long()] - rd).abs().max()) < 2e-6 * float(rd.abs().max())
    assert torch.equal((Gn[pick.long()] == 0), (Gd == 0))
    loader = M.ResidentLoader(frames, sz, 4)
    a.update_footprints(loader, 4, sz, gamma_c=0, iter_c=5, return_dense=False)
    b.update_footprints(loader, 4, sz, gamma_c=0, iter_c=5, return_dense=False)
    assert torch.isfinite(a.C).all()
    rel = (a.C - b.C).abs() / (b.C.abs() + 1e-6)
    assert float(rel.max()) < 1e-4, float(rel.max())


@pytest.mark.parametrize("passes", ["auto", "2"])
def test_full_size_neuron_list_path(M, passes, monkeypatch):
    """The bench's default path at its geometry (512x512, K=100): K3n, the list reconstruction and K4 on the slot
    tables, through the same size-independent properties -- identity warp => A^T A; integer shifts => Gram of the
    shifted footprints; S = A.C against a float64 product; the fused update_footprints equals the two-step one bit
    for bit; a full sweep through the model leaves finite, non-negative traces."""
    if passes != "auto":   # both launch forms of K3n on this problem (DNMF_LISTS_PASSES, warp_gram_lists.hip)
        monkeypatch.setenv("DNMF_LISTS_PASSES", passes)
    from dnmf_amd import ops
    torch.manual_seed(0)
    sz, K, T = [512, 512, 1], 100, 8
    pos = torch.rand(K, 3) * torch.tensor([512.0, 512.0, 0.0])
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    dn.verbose = False
    fp = dn.fp
    frames = torch.rand(T, 512 * 512, device="cuda")
    with torch.no_grad():
        fp.beta[0, 0, 1] = 3.0       # frame 1: shift +3 px in x
        fp.beta[0, 1, 2] = -2.0      # frame 2: shift -2 px in y
    ly = fp.packed_lists()
    assert ly["boxfrac"] < 2 and ly["nbr"] is not None
    G, r, _ = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames)
    A2 = fp.A.reshape(-1, K).double()
    G0 = A2.T @ A2
    scale = float(G0.abs().max())
    for t in (0, 3, 7):
        assert float((G[t].double() - G0).abs().max()) < 2e-5 * scale
        rt = A2.T @ frames[t].double()
        assert float((r[t].double() - rt).abs().max()) < 2e-5 * float(rt.abs().max())
    A3 = fp.A.reshape(512, 512, K)
    sh = torch.zeros_like(A3)
    sh[:-3] = A3[3:]
    S2 = sh.reshape(-1, K).double()
    assert float((G[1].double() - S2.T @ S2).abs().max()) < 2e-5 * scale
    sh = torch.zeros_like(A3)
    sh[:, 2:] = A3[:, :-2]
    S2 = sh.reshape(-1, K).double()
    assert float((G[2].double() - S2.T @ S2).abs().max()) < 2e-5 * scale
    assert torch.equal(G, G.transpose(1, 2))
    # reconstruction image
    C = torch.rand(K, T, device="cuda")
    S = ops.recon_image_lists(ly, K, sz, C, list(range(T)))
    want = (A2 @ C.double()).T
    got = ops.halo_interior(S, sz).reshape(T, -1)
    assert float((got.double() - want).abs().max()) < 1e-5 * float(want.abs().max())
    assert border_is_zero(S, sz)
    # fused update_footprints == Gram, then K4
    dn.C = C.clone()
    test = M.ResidentLoader(frames, sz, 4)
    dn.update_footprints(test, 4, sz, gamma_c=0, iter_c=50, return_dense=False)
    two_step = ops.mu_temporal(G, r, C.clone(), 50, nbr=ly["nbr"])
    assert torch.equal(dn.C, two_step)
    # a sweep through the model
    opt = torch.optim.Adam([fp.beta], lr=1e-5)
    train = M.ResidentLoader(frames, sz, 4, shuffle=True, generator=torch.Generator().manual_seed(0))
    dn.update_motion(train, opt, gamma=1, epochs=1)
    dn.update_footprints(test, 4, sz, gamma_c=0, iter_c=50, return_dense=False)
    assert bool((dn.C >= 0).all()) and bool(torch.isfinite(dn.C).all()) and bool(torch.isfinite(fp.beta).all())


def test_G8_device_simulator(M):
    """The GPU render loop + normalisation against the reference video (noise passed in: the reference
    draws it from the CPU generator)."""
    from dnmf_amd.WUtils import Simulator as S
    g = golden("G8_simulator")
    par = {"sigma": [5, 5, .01], "ls": [10, 10, 10]}
    X, Y, Z = g["sz"].tolist()
    for snr, key in ((-120, "video"), (-20, "video_noisy")):
        torch.manual_seed(0)
        np.random.seed(0)
        noise = np.sqrt(10 ** (snr / 10)) * torch.distributions.normal.Normal(0, 1).sample(np.array([X, Y, Z, 6]))
        np.random.seed(0)
        frames, positions, traces = S.generate_video_resident(3, 6, [X, Y, Z], 3, .2, snr, par, noise=noise)
        got = frames.view(6, X, Y, Z).permute(1, 2, 3, 0).cpu().numpy()
        np.testing.assert_allclose(got, g[key], rtol=5e-6, atol=1e-9)
        np.testing.assert_array_equal(traces, g["traces"])
    # a slice of the T axis is the same frames
    np.random.seed(0)
    part, _, _ = S.generate_video_resident(3, 6, [X, Y, Z], 3, .2, -120, par, noise=torch.zeros(6, X * Y * Z), t0=2, t1=5)
    np.random.seed(0)
    full, _, _ = S.generate_video_resident(3, 6, [X, Y, Z], 3, .2, -120, par, noise=torch.zeros(6, X * Y * Z))
    # different normalisers (slice-local vs global) -> compare up to scale
    np.testing.assert_allclose((part / part.max()).cpu().numpy(), (full[2:5] / full[2:5].max()).cpu().numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("T,bs", [(16, 4), (14, 4)])
def test_fused_motion_epoch_equals_stepwise_adam(M, O, T, bs):
    """The per-column evaluation of an epoch of Adam steps (dnmf_adam_epoch + one K2 launch) against
    torch.optim.Adam stepped mini-batch by mini-batch, over several shuffled epochs (moment history, coasting,
    ragged last batch)."""
    torch.manual_seed(3)
    rng = np.random.RandomState(3)
    sz, K = [24, 20, 2], 6
    pos = rng.rand(K, 3) * np.array(sz)
    frames = torch.rand(T, sz[0] * sz[1] * sz[2], device="cuda")
    C0 = torch.rand(K, T)
    # start off the identity: from the identity the first Adam step moves every coefficient by exactly +-lr,
    # which puts whole curves of voxels exactly ON the source lattice, where the gradient of the trilinear
    # gather is discontinuous and one ulp in beta decides (true of the reference as well)
    jitter = (torch.randn(10, 3, T) * torch.tensor([0.3, 3e-3, 3e-3, 3e-3, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4])[:, None, None])
    res = []
    for fused in (False, True):
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.verbose, dn.fused_motion = False, fused
        dn.C = C0.to("cuda")
        with torch.no_grad():
            dn.fp.beta += jitter.to("cuda")
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
        loader = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(5))
        dn.update_motion(loader, opt, gamma=1, epochs=3)
        dn.update_motion(loader, opt, gamma=1, epochs=2)
        st = opt.state[dn.fp.beta]
        res.append((dn.fp.beta.detach().cpu().numpy(), st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy(),
                    float(st["step"])))
    ident = O.identity_beta(T) + jitter.numpy()
    (b0, m0, v0, s0), (b1, m1, v1, s1) = res
    assert s0 == s1 == 5 * ((T + bs - 1) // bs)
    disp = np.abs(b0 - ident).max()
    np.testing.assert_allclose(b1 - ident, b0 - ident, rtol=0, atol=2e-4 * disp)
    np.testing.assert_allclose(m1, m0, rtol=1e-3, atol=1e-5 * np.abs(m0).max())
    np.testing.assert_allclose(v1, v0, rtol=1e-3, atol=1e-5 * np.abs(v0).max())


@pytest.mark.parametrize("fused", [True, False])
def test_virtual_shards_reproduce_the_single_process_fit(M, fused):
    """Two T-shards (run one after the other on this GPU, each with its own model, optimiser and loader cut from
    the same global permutation) give bit-identical beta columns and C columns to the un-sharded fit."""
    from dnmf_amd import sharding
    torch.manual_seed(11)
    rng = np.random.RandomState(11)
    sz, K, T, bs = [20, 18, 2], 5, 14, 4
    pos = torch.from_numpy(rng.rand(K, 3) * np.array(sz)).float()
    frames = torch.rand(T, sz[0] * sz[1] * sz[2], device="cuda")
    C0 = torch.rand(K, T)
    jitter = torch.randn(10, 3, T) * torch.tensor([0.3, 3e-3, 3e-3, 3e-3, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4])[:, None, None]

    def run(t0, t1):
        dn = M.DeformableNMF(torch.tensor(sz), K, t1 - t0, positions=pos)
        dn.verbose, dn.fused_motion = False, fused
        dn.C = C0[:, t0:t1].to("cuda").contiguous()
        with torch.no_grad():
            dn.fp.beta += jitter[:, :, t0:t1].to("cuda")
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
        train = M.ResidentLoader(frames[t0:t1], sz, bs, shuffle=True, generator=torch.Generator().manual_seed(9),
                                 t0=t0, T_total=T)
        test = M.ResidentLoader(frames[t0:t1], sz, bs)
        for _ in range(2):
            dn.update_motion(train, opt, gamma=1, epochs=2)
            dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=7, return_dense=False)
        return dn.fp.beta.detach().clone(), dn.C.clone()

    beta_full, C_full = run(0, T)
    for r in range(2):
        t0, t1 = sharding.shard_bounds(T, 2, r)
        beta_s, C_s = run(t0, t1)
        assert torch.equal(beta_s, beta_full[:, :, t0:t1])
        assert torch.equal(C_s, C_full[:, t0:t1])


def test_demo_loop_recovers_traces(M, capsys):
    """The reference demo's flow (stock DataLoader over SimulatedVideoDataset, caller-owned Adam, printed progress)
    through the drop-in import path.  The CPU oracle run on the same flow (2 outer iterations, 2 epochs,
    iter_c=30) ends at a median trace correlation of 0.234 with the simulator's ground truth (min 0.069): the
    reference's own fit is that loose at these settings, and the HIP path lands on the same figure."""
    import subprocess
    import sys
    import os
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "demo_headless.py"), "--outer", "2", "--epochs", "2",
                          "--iter-c", "30"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert "Epoch 1" in out.stdout and "Recon: " in out.stdout and "Reg: " in out.stdout
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("trace correlation")][0]
    assert 0.15 < float(line.split("median")[1]) < 0.35, line


@pytest.mark.parametrize("sz,K,T,sigma", [([64, 48, 1], 30, 5, 1.0), ([70, 50, 1], 100, 4, 0.8), ([40, 33, 2], 20, 3, 1.2),
                                          ([24, 40, 5], 40, 3, 0.9), ([33, 47, 1], 50, 4, 3.0), ([96, 80, 1], 200, 3, 0.7),
                                          ([16, 16, 1], 3, 2, 3.0), ([48, 40, 3], 70, 2, 3.0), ([66000, 6, 1], 12, 2, 1.5)])
@pytest.mark.parametrize("passes", ["auto", "2", "2 long runs", "1 long runs"])
def test_neuron_list_gram_equals_dense(M, O, sz, K, T, sigma, passes, monkeypatch):
    """K3n against K3 on the same inputs: narrow footprints (short lists, empty tiles), the reference's sigma = 3 on
    small volumes (every neuron listed everywhere: groups of 4 and the cross-group path), 2-D and 3-D tiles, ragged
    volume edges, K up to 200 (four list words), warps that push part of the volume outside, an axis longer than
    65536 (IEEE division instead of the shortcut).  Sums agree to fp32
    summation order; the pattern of exact zeros of G is the dense kernel's; two launches agree bitwise."""
    if passes != "auto":   # both launch forms of K3n on this problem (DNMF_LISTS_PASSES, warp_gram_lists.hip)
        monkeypatch.setenv("DNMF_LISTS_PASSES", passes[0])
        if "long runs" in passes:   # one wave walks the whole frame: runs of tiles with one and the same list
            monkeypatch.setenv("DNMF_LISTS_CHUNKS", "1")
    from dnmf_amd import ops
    rng = np.random.RandomState(K + sz[0])
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, sigma))
    A[A < 1e-6] = 0          # compact support, as fp32 underflow gives at the reference's scale
    A[..., K - 1] = 0        # one neuron without any non-zero
    beta = O.identity_beta(T)
    sc = min(1.0, 96.0 / max(sz))   # keep the displacement of the far end of a long axis at a few voxels
    beta += (rng.randn(10, 3, T) * np.array([2.0] + [2e-2 * sc] * 3 + [3e-4 * sc * sc] * 6)[:, None, None]).astype(np.float32)
    if sz[2] == 1:
        beta[:, 2] = O.identity_beta(T)[:, 2]
    fp = make_fp(M, sz, K, T, pos, beta=beta, A=A)
    frames = torch.rand(T, int(np.prod(sz)), device="cuda")
    ly = fp.packed_lists()
    bb = ly["bbox"].cpu().numpy()
    nz = A.reshape(-1, K) != 0
    lat = O.voxel_lattice(sz).reshape(-1, 3)
    for k in (0, K // 2, K - 1):
        if nz[:, k].any():
            ref = np.stack([lat[nz[:, k]].min(0), lat[nz[:, k]].max(0)], 1).reshape(-1)
            assert np.array_equal(bb[k], ref.astype(np.int32))
        else:
            assert all(bb[k, 2 * d] > bb[k, 2 * d + 1] for d in range(3))
    At = ly["At"]
    np.testing.assert_array_equal(ops.halo_interior(At, sz).reshape(K, -1).cpu().numpy(), A.reshape(-1, K).T)
    assert border_is_zero(At, sz)
    if K < 128:
        Gd, rd, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), None, frames)
    else:
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.fp, dn.gram_kernel = fp, 'dense'
        Gd, rd = dn._gram_rhs(frames, torch.arange(T, dtype=torch.int32, device="cuda"))
    Gn, rn, ws = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames)
    assert float((Gn - Gd).abs().max()) < 2e-6 * float(Gd.abs().max())
    assert float((rn - rd).abs().max()) < 2e-6 * float(rd.abs().max())
    assert torch.equal(Gn, Gn.transpose(1, 2))
    assert torch.equal(Gn == 0, Gd == 0) or float(Gd[(Gn == 0) != (Gd == 0)].abs().max()) < 1e-30
    G2, r2, _ = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames, workspace=ws)
    assert torch.equal(G2, Gn) and torch.equal(r2, rn)
    # K4 restricted to the pattern of G gives bit-identical traces
    if ly["nbr"] is not None:
        C0 = torch.rand(K, T, device="cuda")
        pat = (ly["pair_slot"] != ly["nslot"] - 1)
        assert bool((Gn[:, ~pat] == 0).all())
        assert all(bool(pat[k, ly["nbr"][k].long()].sum() == pat[k].sum()) for k in range(K))
        assert torch.equal(ops.mu_temporal(Gn, rn, C0.clone(), 9, nbr=ly["nbr"]), ops.mu_temporal(Gn, rn, C0.clone(), 9))
        # ... and so does K4 reading the slot tables directly (no dense G in between)
        _, _, ws2 = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames, finish=False)
        assert torch.equal(ops.mu_temporal_slots(ly, ws2, sz, C0.clone(), 9), ops.mu_temporal(Gn, rn, C0.clone(), 9))
    # a subset of frames in another order, through the model switch
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
    dn.fp, dn.gram_kernel = fp, 'lists'
    order = torch.tensor([T - 1, 0], dtype=torch.int32, device="cuda")
    G3, r3 = dn._gram_rhs(frames[order.long()].contiguous(), order)
    assert torch.equal(G3, Gn[order.long()]) and torch.equal(r3, rn[order.long()])


def test_recon_image_from_lists_long_frame_runs(M, O):
    """The list reconstruction at a frame count where a wave sweeps several runs of 64 frames (the bench's regime: a
    wave takes ~290 frames of a tile at 512x512x4000; the small cases above give it 16): 200x200, 6000 frames, against a
    float64 product on the device."""
    from dnmf_amd import ops
    rng = np.random.RandomState(11)
    sz, K, T = [200, 200, 1], 12, 6000
    A = O.gaussian_footprints(sz, rng.rand(K, 3) * np.array(sz), np.full(K, 2.0))
    A[A < 1e-6] = 0
    P = int(np.prod(sz))
    C = torch.rand(K, T, device="cuda")
    ly = ops.pack_footprints_lists(dev(A), sz)
    times = torch.randperm(T, device="cuda").to(torch.int32)
    S = ops.recon_image_lists(ly, K, sz, C, times)
    ref = (dev(A).reshape(P, K).double() @ C[:, times.long()].double()).T
    got = ops.halo_interior(S, sz).reshape(T, P)
    assert float((got.double() - ref).abs().max()) <= 1e-6 + 1e-5 * float(ref.abs().max())
    assert border_is_zero(S, sz)


@pytest.mark.parametrize("sz,K,sigma", [([64, 48, 1], 30, 1.0), ([70, 130, 1], 100, 0.8), ([40, 33, 2], 20, 1.2),
                                        ([24, 40, 5], 40, 0.9), ([33, 47, 1], 50, 3.0), ([96, 80, 1], 200, 0.7),
                                        ([5, 7, 3], 3, 3.0)])
def test_recon_image_from_lists(M, O, sz, K, sigma):
    """S = A.C from the neuron lists against float64 on the host: short lists, lists longer than the eight register
    slots (read-modify-write of S), ragged tiles, 3-D volumes, K = 200, a frame subset in another order, a padded
    row stride."""
    from dnmf_amd import ops
    rng = np.random.RandomState(K)
    T = 70
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, sigma))
    A[A < 1e-6] = 0
    A[..., 0] = 0
    C = rng.rand(K, T).astype(np.float32)
    P = int(np.prod(sz))
    ly = ops.pack_footprints_lists(dev(A), sz)
    times = rng.permutation(T)[:67]
    Pp = ops.halo_voxels(sz)
    lds = Pp + 8
    out = torch.full((len(times), lds), -1.0, device="cuda")
    S = ops.recon_image_lists(ly, K, sz, dev(C), times, out=out)
    ref = A.reshape(P, K).astype(np.float64) @ C[:, times].astype(np.float64)
    got = ops.halo_interior(S, sz).reshape(len(times), P)
    np.testing.assert_allclose(got.cpu().numpy(), ref.T, rtol=1e-5, atol=1e-6)
    assert bool((S[:, Pp:] == -1.0).all())                      # beyond an image: untouched
    assert border_is_zero(S[:, :Pp], sz)                        # the border of every image: rewritten as zeros


@pytest.mark.parametrize("passes", ["auto", "2"])
def test_neuron_list_gram_under_strong_warps(M, O, passes, monkeypatch):
    """K3n's tile lists come from the taps its voxels actually have, whatever the warp does.  Forty frames whose warps
    range from mild to violent (shifts up to the volume size, shear and scale of order one, quadratic terms that bend
    the volume by tens of voxels, one frame with non-finite coefficients): the result must still be the dense
    kernel's."""
    if passes != "auto":   # both launch forms of K3n on this problem (DNMF_LISTS_PASSES, warp_gram_lists.hip)
        monkeypatch.setenv("DNMF_LISTS_PASSES", passes)
    from dnmf_amd import ops
    rng = np.random.RandomState(123)
    sz, K, T = [112, 96, 1], 60, 40
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, 1.0))
    A[A < 1e-6] = 0
    beta = O.identity_beta(T)
    amp = np.logspace(-3, 1, T)                                  # per-frame strength
    base = np.array([20.0, 0.3, 0.3, 0.3, 3e-3, 3e-3, 3e-3, 3e-3, 3e-3, 3e-3])
    beta += (rng.randn(10, 3, T) * base[:, None, None] * amp[None, None, :]).astype(np.float32)
    beta[:, 2] = O.identity_beta(T)[:, 2]
    beta[4, 0, T - 1] = np.inf                                    # a frame without any finite bound
    fp = make_fp(M, sz, K, T, pos, beta=beta, A=A)
    frames = torch.rand(T, int(np.prod(sz)), device="cuda")
    Gd, rd, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), None, frames)
    Gn, rn, _ = ops.warp_gram_rhs_lists(fp.packed_lists(), K, sz, fp.beta.detach(), None, frames)
    fin = torch.isfinite(Gd).all(dim=(1, 2)) & torch.isfinite(rd).all(dim=1)
    assert int(fin.sum()) >= T - 1
    for t in range(T):
        if not bool(fin[t]):
            continue
        scale = max(float(Gd[t].abs().max()), 1e-30)
        assert float((Gn[t] - Gd[t]).abs().max()) <= 2e-6 * scale, t
        assert float((rn[t] - rd[t]).abs().max()) <= 2e-6 * max(float(rd[t].abs().max()), 1e-30), t
    assert int((Gd.abs().amax(dim=(1, 2)) > 0).sum()) > T // 2   # most frames still see footprints


def test_adam_epoch_against_torch_steps(M):
    """dnmf_adam_epoch (gradient step literal, zero-gradient runs in closed form) against torch.optim.Adam stepped 1500
    times per epoch with the gradient of each column injected at its own step: two epochs, so the second starts from
    moments the first left behind; columns without a mini-batch coast all the way.  torch in float64 is the yardstick:
    the fused epoch must sit within a few fp32 ulps of beta of it (4e-7 + 2e-6 of the displacement), while torch's own
    fp32 run is farther away (thousands of ~1e-5 increments rounded into an fp32 parameter one by one)."""
    from dnmf_amd import ops
    torch.manual_seed(3)
    T, nsteps, lr = 96, 1500, 1e-3
    beta = (torch.randn(10, 3, T, device="cuda") * 0.1 + 1.0).contiguous()
    ref = beta.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=lr)
    ref64 = beta.double().requires_grad_(True)
    opt64 = torch.optim.Adam([ref64], lr=lr)
    m = torch.zeros_like(beta)
    v = torch.zeros_like(beta)
    beta0 = beta.clone()
    for epoch in range(2):
        gen = torch.Generator().manual_seed(epoch)
        fs = torch.randint(0, nsteps, (T,), generator=gen, dtype=torch.int32)
        fs[::7] = -1                                   # frames in no mini-batch of this epoch
        fs[1], fs[2] = 0, nsteps - 1                   # first and last step
        order = torch.argsort(fs, stable=True).to(torch.int32)
        ops.adam_epoch(beta, None, m, v, epoch * nsteps, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=0, order=order)
        grad = torch.randn(10, 3, T, device="cuda") * 1e-2
        grad[:, :, fs.cuda() < 0] = 0
        ops.adam_epoch(beta, grad, m, v, epoch * nsteps, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=1, order=order)
        by_step = {}
        for t in range(T):
            by_step.setdefault(int(fs[t]), []).append(t)
        for s in range(nsteps):
            g = torch.zeros_like(beta)
            for t in by_step.get(s, ()):
                g[:, :, t] = grad[:, :, t]
            ref.grad = g
            opt.step()
            ref64.grad = g.double()
            opt64.step()
    disp = float((ref64.detach() - beta0).abs().max())
    assert disp > 50 * lr
    err = float((beta - ref64.detach()).abs().max())
    err_torch32 = float((ref.detach() - ref64.detach()).abs().max())
    assert err < 4e-7 + 2e-6 * disp, (err, err_torch32, disp)
    assert float((beta - ref.detach()).abs().max()) < 4e-7 + 2e-6 * disp + err_torch32
    st = opt64.state[ref64]
    np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].cpu().numpy(), rtol=1e-4, atol=1e-12)
    np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].cpu().numpy(), rtol=1e-4, atol=1e-20)
    # the same epoch without the lane order gives the same numbers
    b2, m2, v2 = beta0.clone(), torch.zeros_like(beta), torch.zeros_like(beta)
    b3, m3, v3 = beta0.clone(), torch.zeros_like(beta), torch.zeros_like(beta)
    fs = torch.randint(0, nsteps, (T,), generator=torch.Generator().manual_seed(9), dtype=torch.int32)
    grad = torch.randn(10, 3, T, device="cuda") * 1e-2
    for (b, mm, vv), od in (((b2, m2, v2), None), ((b3, m3, v3), torch.argsort(fs, stable=True).to(torch.int32))):
        ops.adam_epoch(b, None, mm, vv, 0, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=0, order=od)
        ops.adam_epoch(b, grad, mm, vv, 0, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=1, order=od)
    assert torch.equal(b2, b3) and torch.equal(m2, m3) and torch.equal(v2, v3)


@pytest.mark.parametrize("variant", ["table", "static"])
@pytest.mark.parametrize("sz,K,T,sigma", [([96, 80, 1], 40, 5, 1.0), ([64, 48, 2], 100, 3, 0.7), ([40, 36, 3], 20, 4, 3.0),
                                          ([33, 47, 1], 7, 6, 0.8), ([32, 32, 1], 100, 3, 3.0), ([48, 40, 2], 80, 3, 1.6)])
def test_sparse_gram_equals_dense(M, O, sz, K, T, sigma, variant):
    """K3s (products with an exact zero skipped) against K3 and against the float64 oracle: footprints narrow
    enough to underflow to exact zeros (sigma <= 1), dense cases where nothing can be skipped (for the table variant
    that is the path with more than four active blocks per half pass), and both kernels (local block table / all
    tiles in registers)."""
    from dnmf_amd import ops
    rng = np.random.RandomState(K)
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, sigma))
    beta = O.identity_beta(T)
    beta += (rng.randn(10, 3, T) * np.array([1.5, 1e-2, 1e-2, 1e-2, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4, 2e-4])[:, None, None]
             ).astype(np.float32)
    if sz[2] == 1:
        beta[:, 2] = O.identity_beta(T)[:, 2]
    video = rng.rand(*sz, T).astype(np.float32)
    fp = make_fp(M, sz, K, T, pos, beta=beta, A=A)
    frames = dev(np.moveaxis(video, -1, 0)).reshape(T, -1)
    sp = fp.packed_sparse()
    assert sorted(sp["order"].tolist()) == list(range(K))
    if sigma <= 1.0:
        assert sp["occupancy"] < 0.6 and float((fp.A == 0).float().mean()) > 0.5
    Gd, rd, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), list(range(T)), frames)
    Gs, rs, _ = ops.warp_gram_rhs_sparse(sp["Aps"], K, sp["order"], sp["row_mask"], sz, fp.beta.detach(), list(range(T)),
                                         frames, variant=variant)
    # same products, same fp32 arithmetic; only the grouping of the partial sums differs
    gs, rsc = float(Gd.abs().max()), float(rd.abs().max())
    assert float((Gs - Gd).abs().max()) < 2e-6 * gs and float((rs - rd).abs().max()) < 2e-6 * rsc
    assert torch.equal(Gs, Gs.transpose(1, 2))
    lat = O.voxel_lattice(sz)
    _, A_t, _, _ = O.forward(A, O.quadratic_basis(lat), beta, sz, list(range(T)), np.zeros((K, T), np.float32),
                             O.trilinear_sample_torch)
    Gref, rref = O.gram_rhs(np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0]), video.astype(np.float64))
    np.testing.assert_allclose(Gs.cpu().numpy(), np.moveaxis(Gref, 2, 0), rtol=2e-5, atol=2e-5 * np.abs(Gref).max())
    np.testing.assert_allclose(rs.cpu().numpy(), rref.T, rtol=2e-5, atol=2e-5 * np.abs(rref).max())


def test_sparse_gram_full_size(M):
    """512x512, K=100, the bench geometry: K3s == K3 on every frame."""
    from dnmf_amd import ops
    torch.manual_seed(0)
    sz, K, T = [512, 512, 1], 100, 8
    pos = torch.rand(K, 3) * torch.tensor([512.0, 512.0, 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    with torch.no_grad():
        scale = torch.tensor([2.0, 2e-3, 2e-3, 0, 2e-6, 2e-6, 0, 2e-6, 0, 0], device="cuda")
        fp.beta += scale[:, None, None] * torch.randn_like(fp.beta)
        fp.beta[:, 2] = 0
        fp.beta[3, 2] = 1
    frames = torch.rand(T, 512 * 512, device="cuda")
    sp = fp.packed_sparse()
    Gd, rd, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), None, frames)
    for variant in ("table", "static"):
        Gs, rs, _ = ops.warp_gram_rhs_sparse(sp["Aps"], K, sp["order"], sp["row_mask"], sz, fp.beta.detach(), None, frames,
                                             variant=variant)
        assert float((Gs - Gd).abs().max()) < 2e-6 * float(Gd.abs().max())
        assert float((rs - rd).abs().max()) < 2e-6 * float(rd.abs().max())
    assert 0.05 < sp["occupancy"] < 0.5


@pytest.mark.parametrize("K,gram", [(130, "dense"), (200, "dense"), (200, "sparse"), (129, "sparse")])
def test_more_than_127_neurons(M, O, K, gram):
    """BASELINE config 5's K=200: Gram by pairs of neuron groups (K3 on groups of 56, K3s on groups of 64 along the
    Z-order curve), recon image by groups, through the fit steps."""
    rng = np.random.RandomState(K)
    sz, T, bs = [28, 24, 2], 6, 3
    pos = rng.rand(K, 3) * np.array(sz)
    video = np.maximum(rng.rand(*sz, T).astype(np.float32) - 0.2, 0)
    C0 = rng.rand(K, T).astype(np.float32)
    beta0 = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([0.5, 5e-3, 5e-3, 5e-3, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4]
                                                                  )[:, None, None]).astype(np.float32)
    ref = O.OracleModel(sz, K, T, pos, C0=C0)
    with torch.no_grad():
        ref.beta_param.copy_(torch.from_numpy(beta0))
    ropt = torch.optim.Adam([ref.beta_param], lr=1e-3)
    batches = [list(range(s0, s0 + bs)) for s0 in range(0, T, bs)]
    ref.update_motion(video, batches, ropt, gamma=1, epochs=1)
    ref.update_footprints(video, bs, gamma_c=0, iter_c=6)

    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
    dn.verbose = False
    dn.gram_kernel = gram
    dn.fp.A = dev(ref.A)
    dn.C = dev(C0)
    with torch.no_grad():
        dn.fp.beta.copy_(dev(beta0))
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
    frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(video, 3, 0)))
    loader = [(frames[b], torch.tensor(b)) for b in batches]
    dn.update_motion(loader, opt, gamma=1, epochs=1)
    dn.update_footprints(loader, bs, sz, gamma_c=0, iter_c=6, return_dense=False)
    disp = np.abs(ref.beta - beta0).max()
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - beta0, ref.beta - beta0, rtol=0, atol=2e-3 * disp)
    np.testing.assert_allclose(dn.C.cpu().numpy(), ref.C, rtol=2e-4, atol=1e-7)
    # static update_spatial with K > 128
    A = rng.rand(9, 7, K)
    Cc = rng.rand(K, 11)
    Yi = rng.rand(9, 7, 11)
    np.testing.assert_allclose(M.DeformableNMF.update_spatial(A, Cc, Yi), O.update_spatial(A, Cc, Yi), rtol=2e-5)


@pytest.mark.parametrize("resident", [False, True])
def test_multichannel_channels_are_extra_voxels(M, O, resident):
    """BASELINE config 5's colour channels (no reference semantics): channels share beta and C, so
    the oracle is the reference arithmetic with every voxel sum also running over channels -- gradient of the mean
    over batch x channels x voxels = mean of the per-channel gradients, G and r summed over channels.  Parity by
    construction only."""
    rng = np.random.RandomState(5)
    sz, K, T, bs, NC = [20, 18, 2], 5, 6, 3, 3
    P = int(np.prod(sz))
    pos = rng.rand(K, 3) * np.array(sz)
    colours = (0.2 + rng.rand(NC, K)).astype(np.float32)
    video = np.maximum(rng.rand(NC, *sz, T).astype(np.float32) - 0.2, 0)      # (C,X,Y,Z,T)
    C0 = rng.rand(K, T).astype(np.float32)
    beta0 = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([0.5, 5e-3, 5e-3, 5e-3, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4]
                                                                  )[:, None, None]).astype(np.float32)
    batches = [list(range(s0, s0 + bs)) for s0 in range(0, T, bs)]

    ref = O.OracleModel(sz, K, T, pos, C0=C0)
    A_base = ref.A.copy()
    with torch.no_grad():
        ref.beta_param.copy_(torch.from_numpy(beta0))
    ropt = torch.optim.Adam([ref.beta_param], lr=1e-3)
    for times in batches:
        g = np.zeros((10, 3, T), np.float32)
        for c in range(NC):
            fr = np.moveaxis(video[c][..., times], -1, 0)
            g += O.mse_beta_grad_autograd(A_base * colours[c], ref.basis, ref.beta, sz, times, C0, fr)[1] / NC
        ropt.zero_grad()
        ref.beta_param.grad = torch.from_numpy(g)
        ropt.step()
    G = np.zeros((K, K, T))
    r = np.zeros((K, T))
    for c in range(NC):
        ref.A = (A_base * colours[c]).astype(np.float32)
        A_t, _, Yv = ref.pushforward(video[c], bs)
        Gc, rc = O.gram_rhs(A_t, Yv)
        G += Gc
        r += rc
    C_ref = O.mu_temporal_from_gram(G, r, C0, gamma=None, iters=7).astype(np.float32)

    dn = M.MultiChannelDNMF(torch.tensor(sz), K, T, colours, positions=torch.from_numpy(pos).float())
    dn.verbose = False
    dn.fp.A = dev(A_base)
    dn.C = dev(C0)
    with torch.no_grad():
        dn.fp.beta.copy_(dev(beta0))
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
    frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(video, 4, 0)))  # (T,C,X,Y,Z)
    if resident:
        loader = M.ResidentLoader(frames.reshape(T, -1), sz, bs)
        assert loader.frames_2d().shape == (T, NC * P)
    else:
        loader = [(frames[b], torch.tensor(b)) for b in batches]
    dn.update_motion(loader, opt, gamma=0, epochs=1)
    assert dn.update_footprints(loader, bs, sz, gamma_c=0, iter_c=7) == (None, None, None)
    disp = np.abs(ref.beta - beta0).max()
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - beta0, ref.beta - beta0, rtol=0, atol=2e-3 * disp)
    np.testing.assert_allclose(dn.C.cpu().numpy(), C_ref, rtol=2e-4, atol=1e-7)
    with pytest.raises(ValueError):
        M.MultiChannelDNMF(torch.tensor(sz), K, T, colours[:, :2], positions=torch.from_numpy(pos).float())


def test_config1_like_run_vs_oracle(M, O):
    """BASELINE configs[0] geometry (64x64, K=10, simulator video) for two outer iterations of the demo loop with
    shuffled mini-batches, against the CPU oracle run on the same batch order."""
    torch.manual_seed(0)
    np.random.seed(0)
    sz, K, T, bs = [64, 64, 2], 10, 24, 4
    video, positions, _ = O.generate_video(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    video = np.maximum(video, 0)
    C0 = torch.rand(K, T)
    gen = torch.Generator().manual_seed(3)
    orders = [[torch.randperm(T, generator=gen).tolist() for _ in range(2)] for _ in range(2)]
    ref = O.OracleModel(sz, K, T, positions[:, :, 0], C0=C0.numpy())
    ropt = torch.optim.Adam([ref.beta_param], lr=1e-4)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(positions[:, :, 0]))
    dn.verbose = False
    dn.fp.A = dev(ref.A)
    dn.C = C0.to("cuda")
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-4)
    frames = torch.from_numpy(np.ascontiguousarray(np.moveaxis(video, 3, 0)))
    test = [(frames[s0:s0 + bs], torch.arange(s0, s0 + bs)) for s0 in range(0, T, bs)]
    for outer in range(2):
        for perm in orders[outer]:
            batches = [perm[s0:s0 + bs] for s0 in range(0, T, bs)]
            ref.update_motion(video, batches, ropt, gamma=1, epochs=1)
            dn.update_motion([(frames[b], torch.tensor(b)) for b in batches], opt, gamma=1, epochs=1)
        ref.update_footprints(video, bs, gamma_c=0, iter_c=10)
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=10, return_dense=False)
    ident = O.identity_beta(T)
    disp = np.abs(ref.beta - ident).max()
    # on-lattice ambiguity after the first Adam step (see the fused-epoch test) limits beta to a few 1e-3 of a step
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - ident, ref.beta - ident, rtol=0, atol=1e-2 * disp)
    # At these settings the reference's own multiplicative update runs away for most neurons (the CPU oracle ends
    # with row maxima from 0.5 up to 1e24: c*r/(Gc + 1e-32) with G_kk ~ 0 for footprints at or beyond the border),
    # and a runaway row amplifies any rounding difference.  Rows that stay bounded must agree tightly; runaway rows
    # must run away here too.
    got, want = dn.C.cpu().numpy(), ref.C
    calm = want.max(1) < 10
    assert calm.sum() >= 3
    np.testing.assert_allclose(got[calm], want[calm], rtol=5e-3, atol=1e-6)  # beta of one frame differs by the lattice tie
    assert np.all(got[~calm].max(1) > 10)


def test_resident_dataset_and_loader(M):
    """SimulatedVideoDataset(resident=True) + .loader(): the whole fit without a host copy of the video."""
    torch.manual_seed(0)
    np.random.seed(0)
    sz, K, T = torch.tensor([40, 32, 2]), 6, 16
    ds = M.SimulatedVideoDataset(K=K, T=T, sz=sz, shape_std=3, density=.2, bg_snr=-120, traces='exp', motion='gp',
                                 motion_par={'sigma': [5, 5, .01], 'ls': [10, 10, 10]}, resident=True)
    assert ds.video.is_cuda and tuple(ds.video.shape) == (40, 32, 2, T) and len(ds) == T
    dn = M.DeformableNMF(sz, K, T, positions=ds.positions[:, :, 0])
    dn.verbose = False
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-5)
    dn.update_motion(ds.loader(4, shuffle=True, generator=torch.Generator().manual_seed(0)), opt, gamma=1, epochs=2)
    A_t, Y_i, Y = dn.update_footprints(ds.loader(4), 4, sz, gamma_c=0, iter_c=5)
    assert A_t.shape == (40, 32, 2, K, T) and np.isfinite(dn.C.cpu().numpy()).all()
    np.testing.assert_allclose(Y, np.maximum(ds.video.double().cpu().numpy(), 0), rtol=0, atol=0)


def test_reconstruction_cache_skips_tiles_without_neurons(M):
    """update_motion keeps its reconstruction images between calls; tiles of the image that no neuron's box reaches are
    zeroed by the first call and left alone afterwards (dnmf_recon_image_lists_ex, skip_empty).  The images must equal a
    fresh reconstruction bit for bit on every call -- also after the traces change, and after the FOOTPRINTS change (new
    boxes: the next call writes everything again, including zeros where a neuron used to be)."""
    from dnmf_amd import ops
    torch.manual_seed(4)
    sz, K, T = [96, 80, 1], 12, 9
    pos = torch.rand(K, 3) * torch.tensor([96.0, 80.0, 0.0])
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    dn.C = torch.rand(K, T, device="cuda") + 0.1
    all_t = torch.arange(T, dtype=torch.int32, device="cuda")

    def fresh():
        return dn.fp.recon_image(dn.C, all_t).clone()

    S1 = dn._recon_cache()[0]
    assert torch.equal(S1, fresh())
    assert float((S1 == 0).float().mean()) > 0.1          # there are empty tiles to skip
    dn.C = torch.rand(K, T, device="cuda") + 0.1
    ly = dn.fp.packed_lists()
    S2 = dn._recon_cache()[0]
    assert S2.data_ptr() == S1.data_ptr() and dn._S_zero[0][0] is ly
    assert torch.equal(S2, fresh())
    # move a neuron: footprints change in place -> new lists -> a full write (the old footprint's voxels must become zero)
    with torch.no_grad():
        A = dn.fp.A
        A[..., 0] = torch.roll(A[..., 0], shifts=(17, -11), dims=(0, 1))
    assert dn.fp.packed_lists() is not ly
    S3 = dn._recon_cache()[0]
    assert torch.equal(S3, fresh())
    S4 = dn._recon_cache()[0]
    assert torch.equal(S4, fresh())


def test_footprint_floor_option(M, O):
    """``ExponentialFP.footprint_floor`` (an extension, default 0: every non-zero value is listed): values below the floor
    leave the neuron lists.  With a floor of 1e-12 the Gram data stay within 2e-6 of the exact-support result (the order
    of the sums changes) and within the usual tolerance of the float64 oracle; the lists get shorter."""
    from dnmf_amd import ops
    torch.manual_seed(2)
    sz, K, T = [160, 128, 1], 30, 4
    pos = torch.rand(K, 3) * torch.tensor([160.0, 128.0, 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    with torch.no_grad():
        fp.beta += 1e-3 * torch.randn_like(fp.beta) * torch.tensor([1.0, 1e-2, 1e-2, 0, 1e-4, 1e-4, 0, 1e-4, 0, 0], device="cuda")[:, None, None]
        fp.beta[:, 2] = torch.tensor([0, 0, 0, 1.0, 0, 0, 0, 0, 0, 0], device="cuda")[:, None]
    frames = torch.rand(T, fp.P, device="cuda")
    G0, r0, _ = ops.warp_gram_rhs_lists(fp.packed_lists(), K, sz, fp.beta.detach(), None, frames)
    box0 = fp.packed_lists()["boxfrac"]
    fp.footprint_floor = 1e-12
    ly = fp.packed_lists()
    assert ly["boxfrac"] < 0.6 * box0
    G1, r1, _ = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames)
    for t in range(T):
        assert float((G1[t] - G0[t]).abs().max()) <= 2e-6 * float(G0[t].abs().max())
        assert float((r1[t] - r0[t]).abs().max()) <= 2e-6 * float(r0[t].abs().max())
    A_t = ops.warp_gather(fp.A, fp.beta.detach(), list(range(T)), want_grid=False)[0].cpu().numpy()    # (T,K,X,Y,Z)
    Gref, rref = O.gram_rhs(np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0]),
                            np.moveaxis(frames.cpu().numpy().reshape(T, *sz), 0, 3).astype(np.float64))
    np.testing.assert_allclose(G1.cpu().numpy(), np.moveaxis(Gref, 2, 0), rtol=2e-5, atol=2e-5 * np.abs(Gref).max())
    np.testing.assert_allclose(r1.cpu().numpy(), rref.T, rtol=2e-5, atol=2e-5 * np.abs(rref).max())
    fp.footprint_floor = 0.0
    assert fp.packed_lists()["boxfrac"] == box0
