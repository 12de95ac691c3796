"""GPU parity of the path bench.py times (ResidentLoader -> fused Adam epoch + one K2 launch on the cached
reconstruction images -> K3n -> K4 on the slot tables) and of the BASELINE.json configurations beyond config 1 / 3:

  test_bench_path_*        the bench's chain end to end against fixture G9 / the CPU oracle
  test_config2_*           256x256, K=50 (configs[1]): forward, Gram data of every Gram kernel, temporal update
  test_config5_*           512x512, 3 channels, K=200, bf16 (configs[4]) through size-independent properties
  test_python_log_det_jac  the Python staticmethod against fixture G2's log_det_jac key

Tolerances as in test_gpu_parity.py unless a test says otherwise.
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


def fixed_order_loader(M, frames, sz, bs, epochs):
    """A ResidentLoader whose epochs use the given mini-batches (list per epoch of lists of frame indices) instead of
    its own permutation, so that a run can follow the batch order a fixture or the oracle used."""
    from dnmf_amd import sharding

    class Fixed(M.ResidentLoader):
        def __init__(self):
            super().__init__(frames, sz, bs, shuffle=False)
            self._epochs = [torch.tensor([i for b in ep for i in b]) for ep in epochs]
            self._next = 0

        def epoch_plan(self):
            perm = self._epochs[self._next % len(self._epochs)]
            self._next += 1
            assert perm.numel() == self.T_total and sorted(perm.tolist()) == list(range(self.T_total))
            return sharding.plan_epoch(perm, self.batch_size, self.t0, self.t0 + self.T)

    return Fixed()


class Calls:
    """Counts the calls of functions of dnmf_amd.ops while a block runs (which kernels a path really took)."""

    def __init__(self, monkeypatch, *names):
        from dnmf_amd import ops
        self.n = {k: 0 for k in names}
        for k in names:
            orig = getattr(ops, k)

            def wrapped(*a, _orig=orig, _k=k, **kw):
                self.n[_k] += 1
                return _orig(*a, **kw)

            monkeypatch.setattr(ops, k, wrapped)


@pytest.mark.parametrize("label", ["lr1e-5_ordered", "lr1e-3_ordered", "lr1e-3_shuffled"])
@pytest.mark.parametrize("passes", ["auto", "2"])
def test_bench_path_vs_fixture_G9(M, O, monkeypatch, label, passes):
    """Fixture G9 (one outer iteration of the reference's demo loop, produced by the reference itself) through the
    bench's path: resident loader, fused Adam epoch (one K2 launch per group of equal mini-batch size on the cached
    reconstruction images), neuron-list Gram kernel; same tolerances as the list-loader test of the same fixture."""
    if passes != "auto":   # both launch forms of K3n on this problem (DNMF_LISTS_PASSES, warp_gram_lists.hip)
        monkeypatch.setenv("DNMF_LISTS_PASSES", passes)
    g = golden("G9_loop")
    lr, epochs, shuffled, bs = g[label + "_cfg"]
    bs = int(bs)
    sz = [int(s) for s in g["sz"]]
    A = O.gaussian_footprints(g["sz"], g["positions"], np.full(4, 3.0))
    dn = M.DeformableNMF(torch.from_numpy(g["sz"]), 4, 8, positions=torch.from_numpy(g["positions"]))
    dn.verbose = False
    dn.fp.A = dev(A)
    dn.C = dev(g[label + "_C0"])
    assert dn.fused_motion and dn.gram_kernel == 'auto'
    opt = torch.optim.Adam([dn.fp.beta], lr=float(lr))
    frames = dev(np.moveaxis(g["video"], -1, 0)).reshape(8, -1)
    order = [[[i for i in b if i >= 0] for b in ep] for ep in g[label + "_order"].tolist()]
    calls = Calls(monkeypatch, "adam_epoch", "warp_gram_rhs_lists", "recon_image_lists")
    train = fixed_order_loader(M, frames, sz, bs, order)
    dn.update_motion(train, opt, gamma=1, epochs=len(order))
    assert calls.n["adam_epoch"] == 2 * len(order) and calls.n["recon_image_lists"] >= 1   # the fused branch ran
    ident = O.identity_beta(8)
    want = g[label + "_beta_after_motion"]
    scale = np.abs(want - ident).max()
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - ident, want - ident, rtol=0, atol=2e-3 * scale)
    test = M.ResidentLoader(frames, sz, bs)
    dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=5, return_dense=False)
    assert calls.n["warp_gram_rhs_lists"] == 1
    np.testing.assert_allclose(dn.C.cpu().numpy(), g[label + "_C_after_footprints"], rtol=1e-4)


@pytest.mark.parametrize("lr", [1e-5, 1e-4])
def test_bench_path_vs_oracle_config1(M, O, monkeypatch, lr):
    """BASELINE configs[0] geometry (64x64, K=10, simulator video), two outer iterations of the demo loop with
    shuffled mini-batches through the bench's path, against the CPU oracle run on the same batch order -- the
    resident / fused counterpart of test_gpu_parity.py::test_config1_like_run_vs_oracle.  With the demo's Adam step
    (lr = 1e-5, demo.py:42) every trace stays in range and EVERY row is compared at rtol 1e-3; with ten times that step
    the warps drift by voxels within the two iterations, traces whose footprint has lost its neuron run away in the
    reference's own update (oracle and HIP path alike), and only the calm rows can be compared (5e-3)."""
    torch.manual_seed(0)
    np.random.seed(0)
    sz, K, T, bs = [64, 64, 2], 10, 24, 4
    video, positions, _ = O.generate_video(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    video = np.maximum(video, 0)
    C0 = torch.rand(K, T)
    gen = torch.Generator().manual_seed(3)
    orders = [[torch.randperm(T, generator=gen).tolist() for _ in range(2)] for _ in range(2)]
    ref = O.OracleModel(sz, K, T, positions[:, :, 0], C0=C0.numpy())
    ropt = torch.optim.Adam([ref.beta_param], lr=lr)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(positions[:, :, 0]))
    dn.verbose = False
    dn.fp.A = dev(ref.A)
    dn.C = C0.to("cuda")
    opt = torch.optim.Adam([dn.fp.beta], lr=lr)
    frames = dev(np.moveaxis(video, 3, 0)).reshape(T, -1)
    calls = Calls(monkeypatch, "adam_epoch", "warp_gram_rhs_lists")
    test = M.ResidentLoader(frames, sz, bs)
    for outer in range(2):
        epochs = [[perm[s0:s0 + bs] for s0 in range(0, T, bs)] for perm in orders[outer]]
        for batches in epochs:
            ref.update_motion(video, batches, ropt, gamma=1, epochs=1)
        dn.update_motion(fixed_order_loader(M, frames, sz, bs, epochs), opt, gamma=1, epochs=2)
        ref.update_footprints(video, bs, gamma_c=0, iter_c=10)
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=10, return_dense=False)
    assert calls.n["adam_epoch"] == 8 and calls.n["warp_gram_rhs_lists"] == 2
    ident = O.identity_beta(T)
    disp = np.abs(ref.beta - ident).max()
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - ident, ref.beta - ident, rtol=0, atol=1e-2 * disp)
    got, want = dn.C.cpu().numpy(), ref.C
    if lr <= 1e-5:
        assert want.max() < 10
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-6)
        return
    calm = want.max(1) < 10   # see test_config1_like_run_vs_oracle: the reference's own update runs away for some rows
    assert calm.sum() >= 3
    np.testing.assert_allclose(got[calm], want[calm], rtol=5e-3, atol=1e-6)
    assert np.all(got[~calm].max(1) > 10)


@pytest.mark.parametrize("passes", ["auto", "2"])
def test_bench_path_slot_tables_vs_oracle(M, O, monkeypatch, passes):
    """The whole chain of a bench sweep including K4 on the slot tables of K3n (taken when the pattern of G has at
    most 32 columns per row and K >= 8): compact footprints well inside a 2-D volume, warps that start off the
    lattice, one sweep = update_motion(epochs=1, shuffled mini-batches) + update_footprints(iter_c=50, gamma_c=0),
    against the CPU oracle on the same batches.  Off-lattice start and well-conditioned Gram matrices (every
    footprint inside the volume), so the tolerances are the tight ones: beta 2e-3 of the largest displacement,
    C rtol 1e-4."""
    if passes != "auto":   # both launch forms of K3n on this problem (DNMF_LISTS_PASSES, warp_gram_lists.hip)
        monkeypatch.setenv("DNMF_LISTS_PASSES", passes)
    rng = np.random.RandomState(11)
    sz, K, T, bs = [96, 80, 1], 12, 8, 4
    pos = np.stack([12 + rng.rand(K) * 72, 12 + rng.rand(K) * 56, np.zeros(K)], 1).astype(np.float32)
    A = O.gaussian_footprints(sz, pos, np.full(K, 1.3))
    A[A < 1e-7] = 0
    Ctrue = 1 + rng.rand(K, T)
    video = np.einsum("xyzk,kt->xyzt", A.astype(np.float64), Ctrue).astype(np.float32)
    video += 0.02 * rng.rand(*video.shape).astype(np.float32)
    C0 = (0.5 + rng.rand(K, T)).astype(np.float32)
    beta0 = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([0.4, 3e-3, 3e-3, 0, 3e-5, 3e-5, 0, 3e-5, 0, 0])[:, None, None]
                                  ).astype(np.float32)
    beta0[:, 2] = O.identity_beta(T)[:, 2]
    perm = rng.permutation(T).tolist()
    batches = [perm[s0:s0 + bs] for s0 in range(0, T, bs)]

    ref = O.OracleModel(sz, K, T, pos, C0=C0)
    ref.A = A.astype(np.float32)
    with torch.no_grad():
        ref.beta_param.copy_(torch.from_numpy(beta0))
    ropt = torch.optim.Adam([ref.beta_param], lr=1e-3)
    ref.update_motion(video, batches, ropt, gamma=1, epochs=1)
    ref.update_footprints(video, bs, gamma_c=0, iter_c=50)

    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos))
    dn.verbose = False
    dn.fp.A = dev(A)
    dn.C = dev(C0)
    with torch.no_grad():
        dn.fp.beta.copy_(dev(beta0))
    opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
    frames = dev(np.moveaxis(video, 3, 0)).reshape(T, -1)
    calls = Calls(monkeypatch, "adam_epoch", "warp_gram_rhs_lists", "mu_temporal_slots", "recon_image_lists")
    dn.update_motion(fixed_order_loader(M, frames, sz, bs, [batches]), opt, gamma=1, epochs=1)
    dn.update_footprints(M.ResidentLoader(frames, sz, bs), bs, sz, gamma_c=0, iter_c=50, return_dense=False)
    assert calls.n == {"adam_epoch": 2, "warp_gram_rhs_lists": 1, "mu_temporal_slots": 1, "recon_image_lists": 1}, calls.n
    disp = np.abs(ref.beta - beta0).max()
    np.testing.assert_allclose(dn.fp.beta.detach().cpu().numpy() - beta0, ref.beta - beta0, rtol=0, atol=2e-3 * disp)
    np.testing.assert_allclose(dn.C.cpu().numpy(), ref.C, rtol=1e-4, atol=1e-7)


def test_config2_geometry_vs_oracle(M, O):
    """BASELINE configs[1]: 256x256 (Z=1), K=50, simulator video (device simulator, demo.py's parameters); four frames
    under non-trivial warps.  forward (K1 + list reconstruction + K2) against the oracle's forward, the Gram data of
    K3, K3s and K3n against the oracle's float64 contraction, 50 temporal updates against the oracle's."""
    from dnmf_amd import ops
    from dnmf_amd.WUtils import Simulator
    torch.manual_seed(2)
    np.random.seed(2)
    sz, K, T = [256, 256, 1], 50, 4
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    pos = positions[:, :, 0].contiguous()
    rng = np.random.RandomState(2)
    beta = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([1.5, 4e-3, 4e-3, 0, 1.5e-5, 1.5e-5, 0, 1.5e-5, 0, 0])[:, None, None]
                                 ).astype(np.float32)
    beta[:, 2] = O.identity_beta(T)[:, 2]
    C = rng.rand(K, T).astype(np.float32)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    dn.verbose = False
    fp = dn.fp
    with torch.no_grad():
        fp.beta.copy_(dev(beta))
    A = fp.A.cpu().numpy()
    lat = O.voxel_lattice(sz)
    basis = O.quadratic_basis(lat)
    times = list(range(T))
    A_tC, A_t, n, reg = O.forward(A, basis, beta, sz, times, C, O.trilinear_sample_torch)
    gA_tC, gA_t, ggrid, greg = fp(times, torch.from_numpy(C))
    # coordinates reach 255: their fp32 spacing (1.5e-5) times the steepest footprint slope (0.29 per voxel at sigma 3)
    np.testing.assert_allclose(gA_t.cpu().numpy(), A_t, rtol=0, atol=1e-5)
    np.testing.assert_allclose(gA_tC.detach().cpu().numpy(), A_tC, rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(ggrid.cpu().numpy(), n, rtol=0, atol=2e-6)
    np.testing.assert_allclose(greg.cpu().numpy(), reg, rtol=1e-3, atol=1e-7)
    video = np.ascontiguousarray(np.moveaxis(frames.cpu().numpy().reshape(T, *sz), 0, 3))
    loss, grad = O.mse_beta_grad_autograd(A, basis, beta, sz, times, C, np.moveaxis(video, -1, 0))
    gl = torch.nn.functional.mse_loss(gA_tC, frames.view(T, *sz))
    gl.backward()
    np.testing.assert_allclose(float(gl), loss, rtol=1e-5)
    noz = [0, 1, 2, 4, 5, 7]
    np.testing.assert_allclose(fp.beta.grad.cpu().numpy()[noz][:, :2], grad[noz][:, :2], rtol=1e-4,
                               atol=1e-4 * np.abs(grad).max())
    A64 = np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0])
    Gref, rref = O.gram_rhs(A64, video.astype(np.float64))
    Gref, rref = np.moveaxis(Gref, 2, 0), rref.T
    order = torch.arange(T, dtype=torch.int32, device="cuda")
    for kernel in ("dense", "sparse", "lists"):
        dn.gram_kernel = kernel
        G, r = dn._gram_rhs(frames, order)
        np.testing.assert_allclose(G.cpu().numpy(), Gref, rtol=2e-5, atol=2e-5 * np.abs(Gref).max(), err_msg=kernel)
        np.testing.assert_allclose(r.cpu().numpy(), rref, rtol=2e-5, atol=2e-5 * np.abs(rref).max(), err_msg=kernel)
    Cref = O.mu_temporal_from_gram(np.moveaxis(Gref, 0, 2), rref.T, C, None, 50)
    dn.gram_kernel = 'auto'
    dn.C = dev(C)
    dn.update_footprints(M.ResidentLoader(frames, sz, 4), 4, sz, gamma_c=0, iter_c=50, return_dense=False)
    np.testing.assert_allclose(dn.C.cpu().numpy(), Cref, rtol=2e-4, atol=1e-7)


def test_config5_geometry_properties(M):
    """BASELINE configs[4]: three colour channels of a 512x512 volume, K=200, bf16 Gram kernel.  The colour axis has no
    reference semantics: channels are extra voxels that share beta and C.  At the full geometry the oracle
    is out of reach, so the checks are size-independent properties: identity warp => G = sum_c colours_c^2 (x) A^T A
    and r = sum_c A_c^T y_c (float64 products on the GPU); an integer shift => Gram of the shifted footprints; the
    fp32 path (K3n per channel) against those to 2e-5, the bf16 path (K3b by pairs of neuron groups) to the 1e-2
    SURVEY 8(c) states; symmetry; a sweep (motion epoch + 50 temporal updates) leaves finite non-negative traces, the
    bf16 traces within 1e-2 of the fp32 ones, and a lower objective in every frame."""
    torch.manual_seed(5)
    sz, K, T, NC = [512, 512, 1], 200, 4, 3
    P = 512 * 512
    pos = torch.rand(K, 3) * torch.tensor([512.0, 512.0, 0.0])
    colours = 0.3 + torch.rand(NC, K)
    dn = M.MultiChannelDNMF(torch.tensor(sz), K, T, colours, positions=pos)
    dn.verbose = False
    fp = dn.fp
    with torch.no_grad():
        fp.beta[0, 0, 1] = 3.0       # frame 1: shift +3 px in x
    A2 = fp.A.reshape(P, K).double()
    Ctrue = 1 + torch.rand(K, T, device="cuda")
    frames = torch.cat([((A2 * colours[c].cuda().double()) @ Ctrue.double()).T.float() for c in range(NC)], 1)   # (T, NC*P)
    frames += 0.01 * torch.rand_like(frames)
    order = torch.arange(T, dtype=torch.int32, device="cuda")
    G0 = sum((A2 * colours[c].cuda().double()).T @ (A2 * colours[c].cuda().double()) for c in range(NC))
    A3 = fp.A.reshape(512, 512, K)
    sh = torch.zeros_like(A3)
    sh[:-3] = A3[3:]
    S2 = sh.reshape(P, K).double()
    G1 = sum((S2 * colours[c].cuda().double()).T @ (S2 * colours[c].cuda().double()) for c in range(NC))
    r0 = sum((A2 * colours[c].cuda().double()).T @ frames[0, c * P:(c + 1) * P].double() for c in range(NC))
    scale = float(G0.abs().max())
    res = {}
    for kernel, tol in (("auto", 2e-5), ("bf16", 1e-2)):
        dn.gram_kernel = kernel
        G, r = dn._gram_rhs(frames, order)
        assert torch.equal(G, G.transpose(1, 2))
        for t in (0, 2, 3):
            assert float((G[t].double() - G0).abs().max()) < tol * scale, kernel
        assert float((G[1].double() - G1).abs().max()) < tol * scale, kernel
        assert float((r[0].double() - r0).abs().max()) < tol * float(r0.abs().max()), kernel
        res[kernel] = (G, r)
    assert float((res["bf16"][0] - res["auto"][0]).abs().max()) < 1e-2 * scale
    # a sweep through the model on either precision
    C0 = torch.rand(K, T, device="cuda") + 0.5
    out = {}
    for kernel in ("auto", "bf16"):
        dn.gram_kernel = kernel
        dn.C = C0.clone()
        with torch.no_grad():
            fp.beta.copy_(torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].repeat(1, 1, T))
        opt = torch.optim.Adam([fp.beta], lr=1e-5)
        train = M.ResidentLoader(frames, sz, 2, shuffle=True, generator=torch.Generator().manual_seed(0))
        dn.update_motion(train, opt, gamma=1, epochs=1)
        dn.update_footprints(M.ResidentLoader(frames, sz, 2), 2, sz, gamma_c=0, iter_c=50)
        assert bool(torch.isfinite(dn.C).all()) and bool((dn.C >= 0).all())
        assert bool(torch.isfinite(fp.beta).all())
        out[kernel] = dn.C.clone()
    assert float((out["bf16"] - out["auto"]).abs().max()) < 1e-2 * float(out["auto"].abs().max())
    # the multiplicative update cannot increase its objective  c^T G c / 2 - r^T c  (G, r under the fitted warp)
    dn.gram_kernel = "auto"
    G, r = dn._gram_rhs(frames, order)

    def objective(Cm):
        c = Cm.double().T                                     # (T,K)
        return 0.5 * torch.einsum("tk,tkl,tl->t", c, G.double(), c) - (r.double() * c).sum(1)

    assert bool((objective(out["auto"]) < objective(C0)).all())


def test_python_log_det_jac(M):
    """ExponentialFP.log_det_jac (the Python staticmethod the reference exposes, Demix/dNMF.py:107-122) against the
    values the reference produced for fixture G2 at the far and the near corner of the volume."""
    g = golden("G2_forward")
    sz = torch.from_numpy(g["sz"]).float()
    for i, t in enumerate(g["times"].tolist()):
        B = torch.from_numpy(g["beta"][:, :, t])
        got = [float(M.ExponentialFP.log_det_jac(B, sz - 1)), float(M.ExponentialFP.log_det_jac(B, sz * 0))]
        np.testing.assert_allclose(got, g["log_det_jac"][i], rtol=1e-5, atol=1e-7)
        Bc = B.cuda()
        got = [float(M.ExponentialFP.log_det_jac(Bc, (sz - 1).cuda())), float(M.ExponentialFP.log_det_jac(Bc, (sz * 0).cuda()))]
        np.testing.assert_allclose(got, g["log_det_jac"][i], rtol=1e-5, atol=1e-7)


# ---- the footprint update the reference leaves commented out -------------------------------------------
@pytest.mark.parametrize("K", [6, 130])
def test_spatial_step_vs_oracle_and_layout_refresh(M, O, K):
    """DeformableNMF.spatial_step (K5 + K6 on fp.A) against the oracle's update_spatial (reference Demix/dNMF.py:151-160,
    float64) on the flattened voxel axis, with and without the distance penalty D, K beyond one K5 launch (130); then
    the packed copies of A must follow the update: update_footprints after a spatial_step has to equal the same call
    on a fresh model built from the updated footprints (K6 writes through the raw pointer)."""
    rng = np.random.RandomState(K)
    sz, T, bs = [28, 24, 2], 8, 4
    P = int(np.prod(sz))
    pos = rng.rand(K, 3) * np.array(sz)
    frames = torch.rand(T, P, device="cuda")
    C0 = (0.2 + rng.rand(K, T)).astype(np.float32)
    D = rng.rand(*sz, K)
    for use_D, gamma in ((False, None), (True, 0.3)):
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.verbose = False
        dn.C = dev(C0)
        A0 = dn.fp.A.cpu().numpy().astype(np.float64)
        test = M.ResidentLoader(frames, sz, bs)
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=3, return_dense=False)    # builds the packed copies
        C1 = dn.C.cpu().numpy().astype(np.float64)
        A_new = dn.spatial_step(frames, D=D if use_D else None, gamma=gamma)
        Yi = frames.cpu().numpy().astype(np.float64).T.reshape(sz[0], sz[1] * sz[2], T)
        want = O.update_spatial(A0.reshape(sz[0], sz[1] * sz[2], K), C1, Yi,
                                D=D.reshape(sz[0], sz[1] * sz[2], K) if use_D else None, gamma=gamma)
        np.testing.assert_allclose(A_new.cpu().numpy().reshape(want.shape), want, rtol=2e-5, atol=1e-30)
        # the next fit steps see the new footprints
        dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=4, return_dense=False)
        fresh = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        fresh.verbose = False
        fresh.fp.A = A_new.clone()
        fresh.C = dev(C1.astype(np.float32))
        fresh.update_footprints(test, bs, sz, gamma_c=0, iter_c=4, return_dense=False)
        assert torch.equal(dn.C, fresh.C)
        S1 = dn.fp.recon_image(dn.C, torch.arange(T, dtype=torch.int32, device="cuda"))
        S2 = fresh.fp.recon_image(fresh.C, torch.arange(T, dtype=torch.int32, device="cuda"))
        assert torch.equal(S1, S2)


def test_live_spatial_update_in_the_fit_loop(M, O):
    """update_footprints(live_spatial=True) = the temporal updates, then K7 on the frames, then spatial_step with the
    constructor's D -- checked against those three steps done by hand, and the registered frames against the oracle's
    image_iwarp (cKDTree) where the nearest neighbour is unique."""
    from dnmf_amd import ops
    rng = np.random.RandomState(3)
    sz, K, T, bs = [30, 26, 1], 5, 6, 3
    P = int(np.prod(sz))
    pos = np.concatenate([4 + rng.rand(K, 2) * np.array([22, 18]), np.zeros((K, 1))], 1).astype(np.float32)
    frames = torch.rand(T, P, device="cuda")
    beta = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([0.6, 6e-3, 6e-3, 0, 1e-4, 1e-4, 0, 1e-4, 0, 0])[:, None, None]
                                 ).astype(np.float32)
    beta[:, 2] = O.identity_beta(T)[:, 2]

    def model():
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos))
        dn.verbose = False
        dn.C = dev(0.3 + rng.rand(K, T).astype(np.float32) * 0 + 0.5)
        with torch.no_grad():
            dn.fp.beta.copy_(dev(beta))
        return dn

    a, b = model(), model()
    test = M.ResidentLoader(frames, sz, bs)
    a.update_footprints(test, bs, sz, gamma_c=0, gamma_a=0.2, iter_c=5, return_dense=False, live_spatial=True)
    b.update_footprints(test, bs, sz, gamma_c=0, iter_c=5, return_dense=False)
    reg = ops.image_iwarp(frames, None, sz, b.fp.beta.detach(), list(range(T)))
    b.spatial_step(reg, D=b.D, gamma=0.2)
    assert torch.equal(a.C, b.C) and torch.equal(a.fp.A, b.fp.A)
    assert not torch.equal(a.fp.A, model().fp.A)
    # registered frames against scipy's nearest-neighbour interpolation (the reference's image_iwarp)
    lat = O.voxel_lattice(sz)
    basis = O.quadratic_basis(lat)
    _, n = O.poly_grid(basis, beta, sz)
    flow = O.pushforward_flow(n, sz)
    for t in range(T):
        want = O.image_iwarp(frames[t].cpu().numpy().reshape(sz), flow[..., t], lat.astype(np.int64))
        got = reg[t].cpu().numpy().reshape(sz)
        mism = got != want.astype(np.float32)
        if mism.any():   # only at lattice points with two (nearly) equidistant warped voxels
            pts = flow[..., t].reshape(-1, 3).astype(np.float64)
            for q in np.flatnonzero(mism.reshape(-1)):
                d = np.sort(((pts - lat.reshape(-1, 3)[q]) ** 2).sum(1))
                assert d[1] - d[0] < 1e-6, (t, q, d[:3])


# ---- stock torch DataLoader (demo.py:33-35) ------------------------------------------------------------------------
class _HostVideo(torch.utils.data.Dataset):
    """Frames on the host behind the reference's dataset protocol: __getitem__ -> (frame (X,Y,Z), index)."""

    def __init__(self, video):
        self.video = video   # (X,Y,Z,T)

    def __len__(self):
        return self.video.shape[3]

    def __getitem__(self, idx):
        return self.video[:, :, :, idx], idx


@pytest.mark.parametrize("pin", [False, True])
def test_stock_dataloader_is_staged_and_matches_the_resident_path(M, O, monkeypatch, pin):
    """A plain torch DataLoader (shuffled, ragged last mini-batch, optionally pinned) drives the same kernels as the
    resident loader: the staged epoch must reproduce, bit for bit, (i) the resident fused epoch given the same batch
    order and (ii) the step-by-step evaluation from host batches when the fused epoch is off."""
    rng = np.random.RandomState(8)
    sz, K, T, bs = [26, 22, 2], 5, 11, 4
    P = int(np.prod(sz))
    pos = rng.rand(K, 3) * np.array(sz)
    video = torch.from_numpy(rng.rand(*sz, T).astype(np.float32))
    frames = video.permute(3, 0, 1, 2).reshape(T, P).cuda()
    C0 = rng.rand(K, T).astype(np.float32)

    def model(**kw):
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.verbose = False
        dn.C = dev(C0)
        for k, v in kw.items():
            setattr(dn, k, v)
        return dn, torch.optim.Adam([dn.fp.beta], lr=1e-3)

    def loader(seed):
        return torch.utils.data.DataLoader(_HostVideo(video), batch_size=bs, shuffle=True, pin_memory=pin,
                                           generator=torch.Generator().manual_seed(seed))

    # the batch order the DataLoader will serve
    served = [[idx.tolist() for _, idx in loader(5)] for _ in range(1)][0]
    assert [len(b) for b in served] == [4, 4, 3]
    # (i) fused: staged DataLoader == resident loader with that order
    a, opt_a = model()
    calls = Calls(monkeypatch, "adam_epoch")
    a.update_motion(loader(5), opt_a, gamma=1, epochs=1)
    assert calls.n["adam_epoch"] == 2
    b, opt_b = model()
    b.update_motion(fixed_order_loader(M, frames, sz, bs, [served]), opt_b, gamma=1, epochs=1)
    assert torch.equal(a.fp.beta, b.fp.beta)
    a.update_footprints(torch.utils.data.DataLoader(_HostVideo(video), batch_size=bs), bs, sz, gamma_c=0, iter_c=7)
    b.update_footprints(M.ResidentLoader(frames, sz, bs), bs, sz, gamma_c=0, iter_c=7, return_dense=False)
    assert torch.equal(a.C, b.C)
    # (ii) step by step: staged == batch by batch from the host
    c, opt_c = model(fused_motion=False)
    c.update_motion(loader(6), opt_c, gamma=1, epochs=2)
    d, opt_d = model(fused_motion=False, stream_loader=False)
    d.update_motion(loader(6), opt_d, gamma=1, epochs=2)
    assert torch.equal(c.fp.beta, d.fp.beta)
    assert not torch.equal(c.fp.beta, a.fp.beta)


def test_stock_dataloader_over_the_simulated_dataset_fetches_no_sample(M, O, monkeypatch):
    """demo.py:26-38 as it stands: SimulatedVideoDataset + stock DataLoaders.  The fit takes the frames from
    ``dataset.device_frames()`` and draws only the loaders' index batches: no ``__getitem__`` call, the shuffled order
    (global RNG, no generator, as in the demo) is the one an ordinary pass over the loader draws, and beta, C equal the
    host-staged path's bit for bit; the stored video ends clamped like after a pass of ``__getitem__`` calls."""
    rng = np.random.RandomState(9)
    sz, K, T, bs = [24, 20, 2], 4, 14, 4
    pos = rng.rand(K, 3) * np.array(sz)
    video = torch.from_numpy((rng.rand(*sz, T) - 0.1).astype(np.float32))   # some negative values: the clamp matters
    C0 = rng.rand(K, T).astype(np.float32)

    def dataset():
        ds = M.SimulatedVideoDataset.__new__(M.SimulatedVideoDataset)   # the constructor would run the simulator
        ds.video, ds.positions, ds.traces, ds._sz = video.clone(), None, None, list(sz)
        return ds

    def model():
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.verbose = False
        dn.C = dev(C0)
        return dn, torch.optim.Adam([dn.fp.beta], lr=1e-3)

    def run(ds, count):
        torch.manual_seed(77)   # the demo's loaders shuffle from the global generator
        train = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0)
        test = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0)
        dn, opt = model()
        for _ in range(2):
            dn.update_motion(train, opt, gamma=1, epochs=2)
            dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=5, return_dense=False)
        return dn, float(torch.rand(1))   # where the global generator stands afterwards

    served = []
    fast = dataset()
    monkeypatch.setattr(M.SimulatedVideoDataset, "__getitem__", lambda self, i: served.append(i) or (self.video[..., i], i))
    a, rng_a = run(fast, True)
    assert served == []
    assert float(fast.video.min()) == 0.0 and float(video.min()) < 0.0
    monkeypatch.undo()

    class Plain(torch.utils.data.Dataset):   # the same frames behind a dataset without device_frames: host staging
        def __init__(self, v):
            self.video = v

        def __len__(self):
            return self.video.shape[3]

        def __getitem__(self, idx):
            sample = self.video[:, :, :, idx]
            sample[sample < 0] = 0
            return sample, idx

    b, rng_b = run(Plain(video.clone()), False)
    assert torch.equal(a.fp.beta, b.fp.beta) and torch.equal(a.C, b.C)
    assert rng_a == rng_b
    # a write to the stored video renews the device copy
    fast.video[..., 3] += 1.0
    assert torch.equal(fast.device_frames()[3].cpu(), fast.video[..., 3].reshape(-1))


def test_recorded_video_dataset_hands_over_its_frames(M, tmp_path):
    """NeuroPALVideoDataset behind stock DataLoaders: the fit takes ``device_frames()`` and gets the traces and warps of
    the same frames served sample by sample (a dataset class without that method)."""
    from scipy.io import savemat
    rng = np.random.RandomState(3)
    data = rng.rand(24, 20, 20, 9) - 0.05
    pos = 1 + rng.rand(3, 3, 9) * np.array([24, 20, 20])[None, :, None]
    savemat(tmp_path / "data.mat", {"data": data})
    savemat(tmp_path / "traces_n.mat", {"positions": pos, "neuron_names": np.array([["a", "b", "c"]], dtype=object)})

    class SampleBySample(torch.utils.data.Dataset):
        def __init__(self, ds):
            self.ds = ds

        def __len__(self):
            return len(self.ds)

        def __getitem__(self, i):
            return self.ds[i]

    results = []
    for wrap in (False, True):
        ds = M.NeuroPALVideoDataset(str(tmp_path))
        sz, T, K = list(ds.video.shape[:3]), len(ds), ds.positions.shape[0]
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=ds.positions[:, :, 0])
        dn.verbose = False
        dn.C = dev(np.linspace(0.2, 1.0, K * T, dtype=np.float32).reshape(K, T))
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-4)
        served = SampleBySample(ds) if wrap else ds
        dn.update_motion(torch.utils.data.DataLoader(served, batch_size=4, shuffle=True,
                                                     generator=torch.Generator().manual_seed(2)), opt, gamma=1, epochs=2)
        dn.update_footprints(torch.utils.data.DataLoader(served, batch_size=4), 4, sz, gamma_c=0, iter_c=6, return_dense=False)
        results.append((dn.fp.beta.detach().clone(), dn.C.clone(), float(ds.video.min())))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    assert results[0][2] == 0.0 and results[1][2] == 0.0


@pytest.mark.parametrize("sz", [[1, 40, 3], [9, 1, 1], [600, 3, 1]])
def test_static_update_temporal_takes_A_t_as_it_is(M, O, sz):
    """The static update_temporal (reference Demix/dNMF.py:139-149) contracts the A_t it is handed without re-sampling
    it: axes of length one (a trilinear round trip divides 0/0 there) and a long axis (where an fp32 round trip of the
    lattice is off by a few 1e-6 voxels) must match the oracle like any other shape."""
    rng = np.random.RandomState(sum(sz))
    K, T = 7, 3
    A_t = rng.rand(*sz, K, T)
    C = 0.3 + rng.rand(K, T)
    Y = rng.rand(*sz, T)
    for gamma in (None, 1e-2):
        got = M.DeformableNMF.update_temporal(A_t, C, Y, gamma=gamma)
        want = O.update_temporal(A_t, C, Y, gamma=gamma)
        np.testing.assert_allclose(got, want, rtol=2e-5)
    # more than 127 neurons: by pairs of neuron groups (the reference has no limit, Demix/dNMF.py:139-149)
    Kb = 130
    A_t, C, Y = rng.rand(6, 5, 2, Kb, 2), 0.3 + rng.rand(Kb, 2), rng.rand(6, 5, 2, 2)
    np.testing.assert_allclose(M.DeformableNMF.update_temporal(A_t, C, Y), O.update_temporal(A_t, C, Y), rtol=2e-5)


def test_multichannel_spatial_step(M):
    """MultiChannelDNMF.spatial_step (not in the reference: channels are extra voxels that share beta and C, and channel c
    shows neuron k as colours[c,k] A[:,k]): against the update formula in float64, and for ONE channel of colour 1 against
    DeformableNMF.spatial_step bit for bit (dense kernels)."""
    rng = np.random.RandomState(4)
    sz, K, T, NC = [20, 16, 2], 9, 12, 3
    P = int(np.prod(sz))
    pos = torch.from_numpy(rng.rand(K, 3) * np.array(sz)).float()
    colours = torch.from_numpy(0.3 + rng.rand(NC, K)).float()
    frames = torch.rand(T, NC * P, device="cuda")
    C0 = dev(0.2 + rng.rand(K, T))
    D = rng.rand(*sz, K)
    dn = M.MultiChannelDNMF(torch.tensor(sz), K, T, colours, positions=pos)
    dn.C = C0.clone()
    A0 = dn.fp.A.reshape(P, K).double().cpu().numpy()
    got = dn.spatial_step(frames, D=D, gamma=0.2).reshape(P, K).double().cpu().numpy()
    Cn, col = C0.double().cpu().numpy(), colours.double().numpy()
    Y = frames.double().cpu().numpy().reshape(T, NC, P)
    A1 = sum(col[c][None, :] * (Y[:, c].T @ Cn.T) for c in range(NC))
    den = A0 @ ((Cn @ Cn.T) * (col.T @ col)) + 0.2 * D.reshape(P, K) + 1e-32
    np.testing.assert_allclose(got, A0 * A1 / den, rtol=3e-5, atol=1e-30)
    one = M.MultiChannelDNMF(torch.tensor(sz), K, T, torch.ones(1, K), positions=pos)
    one.C = C0.clone()
    ref = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    ref.C = C0.clone()
    ref.spatial_kernel = "dense"
    a = one.spatial_step(frames[:, :P].contiguous(), D=D, gamma=0.2)
    b = ref.spatial_step(frames[:, :P].contiguous(), D=D, gamma=0.2)
    assert torch.equal(a, b)


def test_short_adam_epoch_is_torch_bit_for_bit(M):
    """An epoch of at most 64 mini-batches: dnmf_adam_epoch steps every column through torch's own fp32 sequence (the
    fused multiply-adds of its GPU kernels, tools/adam_probe.py), so beta and both moments must EQUAL those of
    torch.optim.Adam stepped once per mini-batch with the gradients injected at their steps -- two epochs, frames without
    a mini-batch, first and last step."""
    from dnmf_amd import ops
    torch.manual_seed(4)
    T, nsteps, lr = 40, 25, 1e-3
    beta = (torch.randn(10, 3, T, device="cuda") * 0.1 + 1.0).contiguous()
    ref = beta.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=lr)
    m, v = torch.zeros_like(beta), torch.zeros_like(beta)
    for epoch in range(2):
        gen = torch.Generator().manual_seed(epoch)
        fs = torch.randint(0, nsteps, (T,), generator=gen, dtype=torch.int32)
        fs[::7] = -1
        fs[1], fs[2] = 0, nsteps - 1
        ops.adam_epoch(beta, None, m, v, epoch * nsteps, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=0)
        grad = torch.randn(10, 3, T, device="cuda") * 1e-2
        grad[:, :, fs.cuda() < 0] = 0
        ops.adam_epoch(beta, grad, m, v, epoch * nsteps, fs, nsteps, lr, (0.9, 0.999), 1e-8, phase=1)
        for s in range(nsteps):
            g = torch.zeros_like(beta)
            sel = (fs == s).cuda()
            g[:, :, sel] = grad[:, :, sel]
            ref.grad = g
            opt.step()
        st = opt.state[ref]
        assert torch.equal(beta, ref.detach()), float((beta - ref.detach()).abs().max())
        assert torch.equal(m, st["exp_avg"]) and torch.equal(v, st["exp_avg_sq"])


@pytest.mark.parametrize("sz", [[20, 16, 2], [40, 36, 1]])
def test_fused_motion_epochs_equal_the_stepwise_run_bit_for_bit(M, O, sz):
    """update_motion through the fused epoch (per-column Adam kernel, one K2 launch per group of mini-batches) and
    through one K2 launch + one optimizer.step() of torch per mini-batch: the same beta, bit for bit, after three
    shuffled epochs from the identity -- the start at which one ulp of beta decides which lattice cell a voxel falls
    into (the first Adam step moves every coefficient by exactly lr), so "close" would not be good enough."""
    rng = np.random.RandomState(1)
    K, T, bs = 4, 10, 4
    pos = rng.rand(K, 3) * np.array(sz)
    video = torch.from_numpy(rng.rand(T, *sz).astype(np.float32))
    orders = [rng.permutation(T).tolist() for _ in range(3)]
    out = []
    for fused in (True, False):
        torch.manual_seed(0)
        dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
        dn.verbose, dn.fused_motion = False, fused
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
        for perm in orders:
            batches = [perm[s0:s0 + bs] for s0 in range(0, T, bs)]
            dn.update_motion([(video[b], torch.tensor(b)) for b in batches], opt, gamma=1, epochs=1)
        out.append((dn.fp.beta.detach().clone(), opt.state[dn.fp.beta]["exp_avg"].clone(),
                    opt.state[dn.fp.beta]["exp_avg_sq"].clone(), float(opt.state[dn.fp.beta]["step"])))
    assert out[0][3] == out[1][3] == 9.0
    assert torch.equal(out[0][0], out[1][0]), float((out[0][0] - out[1][0]).abs().max())
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][2], out[1][2])
    assert float((out[0][0] - torch.from_numpy(O.identity_beta(T)).cuda()).abs().max()) > 1e-3


def test_motion_grad_in_pieces_is_the_same_gradient(M):
    """dnmf_motion_grad_lists (reconstruction images made `chunk` frames at a time into one small buffer) against
    dnmf_recon_image_lists + dnmf_warp_recon_grad on all frames: identical gradient, losses and reg, also with row ids,
    a piece size that does not divide the number of frames and through the model (motion_chunk)."""
    from dnmf_amd import ops
    torch.manual_seed(2)
    sz, K, T = [70, 52, 1], 9, 23
    pos = torch.rand(K, 3) * torch.tensor([70.0, 52.0, 0.0])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    with torch.no_grad():
        fp.beta += 1e-2 * torch.randn_like(fp.beta) * torch.tensor([50, 1, 1, 0, 1e-2, 1e-2, 0, 1e-2, 0, 0], device="cuda")[:, None, None]
        fp.beta[:, 2] = torch.tensor([0., 0, 0, 1, 0, 0, 0, 0, 0, 0], device="cuda")[:, None]
    C = torch.rand(K, T, device="cuda")
    frames = torch.rand(T + 5, fp.P, device="cuda")
    rows = torch.randperm(T + 5)[:T].to(torch.int32).cuda()
    times = torch.randperm(T).to(torch.int32).cuda()
    ly = fp.packed_lists()
    beta = fp.beta.detach()
    S = ops.recon_image_lists(ly, K, sz, C, times)
    g0 = torch.zeros_like(beta)
    ref = ops.warp_recon_grad(S, None, frames, rows, sz, beta, times, grad=g0, norm_frames=4)
    for chunk in (1, 5, 8, 23, 64):
        g = torch.zeros_like(beta)
        out = ops.motion_grad_lists(ly, K, sz, C, frames, rows, beta, times, g, 4, chunk, want=True)
        assert torch.equal(g, g0), chunk
        assert torch.equal(out["frame_loss"], ref["frame_loss"]) and torch.equal(out["reg"], ref["reg"])
    a = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    b = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    res = []
    for dn, chunk in ((a, 0), (b, 6)):
        dn.verbose, dn.motion_chunk = False, chunk
        dn.C = C.clone()
        opt = torch.optim.Adam([dn.fp.beta], lr=1e-3)
        dn.update_motion(M.ResidentLoader(frames[:T], sz, 4, shuffle=True, generator=torch.Generator().manual_seed(1)), opt,
                         gamma=1, epochs=2)
        res.append(dn.fp.beta.detach().clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("sz", [[64, 48, 2], [37, 30, 1], [21, 9, 3]])
def test_list_form_of_the_footprint_update(M, O, sz):
    """K5 / K6 in their list form (sums only for the (tile, listed neuron) pairs whose box meets the tile) against the dense
    kernels and against the oracle's update_spatial (reference Demix/dNMF.py:151-160) on compact footprints: identical
    zero pattern (a zero stays a zero), the new values equal to the dense path's to the rounding of the frame sums (1e-5)
    and to the oracle's float64 result; with and without D; plane sizes that are and are not multiples of four voxels;
    through a frame-index indirection.  A tile with more than 32 neurons sends 'auto' back to the dense kernels."""
    from dnmf_amd import ops
    rng = np.random.RandomState(sum(sz))
    K, T = 14, 70
    P = int(np.prod(sz))
    pos = rng.rand(K, 3) * np.array(sz)
    A = O.gaussian_footprints(sz, pos, np.full(K, 1.2))
    A[A < 1e-5] = 0
    frames = torch.rand(T + 5, P, device="cuda")
    rows = torch.randperm(T + 5)[:T].to(torch.int32).cuda()
    C0 = (0.2 + rng.rand(K, T)).astype(np.float32)
    D = rng.rand(*sz, K)
    res = {}
    for kernel in ("dense", "lists"):
        for use_D in (False, True):
            dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
            dn.verbose = False
            dn.fp.A = dev(A)
            dn.C = dev(C0)
            dn.spatial_kernel = kernel
            res[kernel, use_D] = dn.spatial_step(frames, D=D if use_D else None, gamma=0.3 if use_D else None, frame_ids=rows,
                                                 times=torch.arange(T, dtype=torch.int32)).cpu().numpy().astype(np.float64)
            if kernel == "lists" and P >= 4096:   # (tiles are padded to 256 voxels: tiny volumes gain nothing)
                assert dn._spatial_buf.numel() < P * K // 2 + K * K          # the exchanged buffer is the compact one
    Yi = frames[rows.long()].cpu().numpy().astype(np.float64).T.reshape(sz[0], sz[1] * sz[2], T)
    for use_D in (False, True):
        want = O.update_spatial(A.astype(np.float64).reshape(sz[0], sz[1] * sz[2], K), C0.astype(np.float64), Yi,
                                D=D.reshape(sz[0], sz[1] * sz[2], K) if use_D else None, gamma=0.3 if use_D else None).reshape(*sz, K)
        got = res["lists", use_D]
        assert np.array_equal(got != 0, A != 0)
        np.testing.assert_allclose(got, res["dense", use_D], rtol=1e-5, atol=1e-30)
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-30)
    # every neuron in one place: the tile holds more than 32 of them
    Kc = 40
    dn = M.DeformableNMF(torch.tensor([32, 32, 2]), Kc, 8, positions=torch.tensor([[16.0, 16.0, 1.0]]).repeat(Kc, 1))
    dn.fp.A = dn.fp.A * (dn.fp.A > 1e-3)
    assert ops.spatial_lists_setup(dn.fp.packed_lists(), Kc, [32, 32, 2])["total"] == -1
    dn.fp.use_lists = False
    dn.spatial_step(torch.rand(8, 2048, device="cuda"))                      # 'auto' -> dense
    dn.spatial_kernel = "lists"
    with pytest.raises(ValueError):
        dn.spatial_step(torch.rand(8, 2048, device="cuda"))
