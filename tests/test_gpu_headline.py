"""The oracle chain at the HEADLINE geometry (BASELINE configs[2]: 512x512, K=100) and at the reference's own depth
(512x512x2): the same recipe as test_gpu_configs.py::test_config2_geometry_vs_oracle, two or three simulator frames under
off-lattice quadratic warps of a few voxels.  Everything size-dependent in the kernels is on at this size and nowhere
below it: the division shortcut per divisor 511, fp32 tap offsets near their 2^24-byte limit, 8 x 32 (8 x 16 x 2) tiles
in rows of 516 (1032) floats, the LDS regions of K3n, both of its launch forms.

  forward      A_t, A_tC, grid, reg                 vs O.forward(..., trilinear_sample_torch)
  gradient     d mse / d beta                        vs O.mse_beta_grad_autograd (torch-CPU autograd)
  Gram data    G, r of K3, K3s, K3n (library's choice, both launches, one chunk per frame)   vs O.gram_rhs in float64
  traces       C after 50 multiplicative updates     vs O.mu_temporal_from_gram

plus K3n against the dense kernel under LARGE quadratic coefficients whose terms cancel (the margin of its tile lists).
The oracle costs 1-6 s per frame at this size: ~1 min per case on the GPU box's host.
"""
import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("Z,T", [(1, 3), (2, 2)])
def test_headline_geometry_vs_oracle(M, O, monkeypatch, Z, T):
    from dnmf_amd import ops
    from dnmf_amd.WUtils import Simulator
    torch.manual_seed(3)
    np.random.seed(3)
    sz, K = [512, 512, Z], 100
    frames, positions, _ = Simulator.generate_video_resident(K, T, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    frames.clamp_(min=0)
    pos = positions[:, :, 0].contiguous()
    rng = np.random.RandomState(3)
    # shifts of ~2 px, linear terms of 0.4 %, quadratic terms worth ~2 px at the far corner; for Z = 2 the z row moves too
    # (a shift of 0.2 slices, a tilt of half a slice across the volume)
    amp = np.array([2.0, 4e-3, 4e-3, 0, 8e-6, 8e-6, 0, 8e-6, 0, 0])
    beta = O.identity_beta(T) + (rng.randn(10, 3, T) * amp[:, None, None]).astype(np.float32)
    beta[:, 2] = O.identity_beta(T)[:, 2]
    if Z > 1:
        beta[0, 2] += (0.2 * rng.randn(T)).astype(np.float32)
        beta[1:3, 2] += (1e-3 * rng.randn(2, T)).astype(np.float32)
    C = (0.5 + rng.rand(K, T)).astype(np.float32)
    dn = M.DeformableNMF(torch.tensor(sz), K, T, positions=pos)
    dn.verbose = False
    fp = dn.fp
    with torch.no_grad():
        fp.beta.copy_(dev(beta))
    A = fp.A.cpu().numpy()
    basis = O.quadratic_basis(O.voxel_lattice(sz))
    times = list(range(T))
    # ---- forward
    A_tC, A_t, n, reg = O.forward(A, basis, beta, sz, times, C, O.trilinear_sample_torch)
    gA_tC, gA_t, ggrid, greg = fp(times, torch.from_numpy(C))
    # coordinates reach 511: their fp32 spacing (3e-5) times the steepest footprint slope (0.29 per voxel at sigma 3)
    np.testing.assert_allclose(gA_t.cpu().numpy(), A_t, rtol=0, atol=2e-5)
    np.testing.assert_allclose(gA_tC.detach().cpu().numpy(), A_tC, rtol=1e-5, atol=4e-5)
    np.testing.assert_allclose(ggrid.cpu().numpy(), n, rtol=0, atol=2e-6)
    np.testing.assert_allclose(greg.cpu().numpy(), reg, rtol=1e-3, atol=1e-7)
    del gA_t, ggrid
    # ---- gradient of the motion step
    video = np.ascontiguousarray(np.moveaxis(frames.cpu().numpy().reshape(T, *sz), 0, 3))
    loss, grad = O.mse_beta_grad_autograd(A, basis, beta, sz, times, C, np.moveaxis(video, -1, 0))
    gl = torch.nn.functional.mse_loss(gA_tC, frames.view(T, *sz))
    gl.backward()
    np.testing.assert_allclose(float(gl), loss, rtol=1e-5)
    ggrad = fp.beta.grad.cpu().numpy()
    if Z == 1:   # the z column and the rows of the terms with z have no meaning for a single slice
        noz = [0, 1, 2, 4, 5, 7]
        ggrad, grad = ggrad[noz][:, :2], grad[noz][:, :2]
    np.testing.assert_allclose(ggrad, grad, rtol=1e-4, atol=1e-4 * np.abs(grad).max())
    # ---- Gram data of every Gram kernel
    A64 = np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0])
    del A_t
    Gref, rref = O.gram_rhs(A64, video.astype(np.float64))
    del A64
    Gref, rref = np.moveaxis(Gref, 2, 0), rref.T
    order = torch.arange(T, dtype=torch.int32, device="cuda")

    def check(label):
        G, r = dn._gram_rhs(frames, order)
        np.testing.assert_allclose(G.cpu().numpy(), Gref, rtol=2e-5, atol=2e-5 * np.abs(Gref).max(), err_msg=label)
        np.testing.assert_allclose(r.cpu().numpy(), rref, rtol=2e-5, atol=2e-5 * np.abs(rref).max(), err_msg=label)

    for kernel in ("dense", "sparse", "lists"):
        dn.gram_kernel = kernel
        check(kernel)
    dn.gram_kernel = "lists"
    monkeypatch.setenv("DNMF_LISTS_PASSES", "2")       # both launches, side stream, separate tables: what the bench runs
    check("lists, two launches")
    monkeypatch.setenv("DNMF_LISTS_CHUNKS", "1")       # one chunk per frame: the long runs of tiles of the bench
    check("lists, two launches, one chunk per frame")
    monkeypatch.setenv("DNMF_LISTS_PASSES", "1")
    check("lists, one launch, one chunk per frame")
    monkeypatch.delenv("DNMF_LISTS_PASSES")
    monkeypatch.delenv("DNMF_LISTS_CHUNKS")
    # ---- 50 temporal updates through update_footprints (K3n + K4 on the slot tables)
    Cref = O.mu_temporal_from_gram(np.moveaxis(Gref, 0, 2), rref.T, C, None, 50)
    dn.gram_kernel = 'auto'
    dn.C = dev(C)
    dn.update_footprints(M.ResidentLoader(frames, sz, 4), 4, sz, gamma_c=0, iter_c=50, return_dense=False)
    np.testing.assert_allclose(dn.C.cpu().numpy(), Cref, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("passes", ["1", "2"])
def test_neuron_list_gram_with_cancelling_quadratic_terms(M, monkeypatch, passes):
    """K3n's tile lists come from interval arithmetic on the ten terms of the coordinate polynomial plus a margin for the
    rounding of either evaluation.  With quadratic coefficients of order 0.1-1 at 512x512 whose terms cancel along a band
    (q_x = x + c x (x - y): the identity on the diagonal), the partial sums reach |c| 512^2 ~ 10^5 voxels and the chain's
    rounding ~0.1 voxel -- more than a fixed margin; the margin therefore grows with the summed term magnitudes
    (common.hpp: tap_range).  The result must be the dense kernel's: same zero pattern, 2e-6 of the largest entry."""
    monkeypatch.setenv("DNMF_LISTS_PASSES", passes)
    from dnmf_amd import ops
    rng = np.random.RandomState(11)
    sz, K, T = [512, 512, 1], 60, 6
    t = rng.rand(K) * 500 + 6                                    # footprints along the diagonal band
    pos = np.stack([t, np.clip(t + rng.randn(K) * 3, 1, 510), np.zeros(K)], 1)
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=torch.from_numpy(pos).float())
    beta = np.zeros((10, 3, T), dtype=np.float32)
    beta[1, 0] = beta[2, 1] = beta[3, 2] = 1.0
    for i, c in enumerate([0.0, 0.03, 0.1, 0.3, 0.6, 1.0]):
        beta[4, 0, i], beta[7, 0, i] = c, -c                     # q_x = x + c x^2 - c x y
        beta[5, 1, i], beta[7, 1, i] = -0.5 * c, 0.5 * c         # q_y = y - c/2 y^2 + c/2 x y
    with torch.no_grad():
        fp.beta.copy_(torch.from_numpy(beta).cuda())
    frames = torch.rand(T, 512 * 512, device="cuda")
    Gd, rd, _ = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), None, frames)
    Gn, rn, _ = ops.warp_gram_rhs_lists(fp.packed_lists(), K, sz, fp.beta.detach(), None, frames)
    assert int((Gd.abs().amax(dim=(1, 2)) > 0).sum()) == T      # every frame still sees footprints
    for i in range(T):
        scale = float(Gd[i].abs().max())
        assert float((Gn[i] - Gd[i]).abs().max()) <= 2e-6 * scale, i
        assert float((rn[i] - rd[i]).abs().max()) <= 2e-6 * float(rd[i].abs().max()), i
        assert torch.equal(Gn[i] != 0, Gd[i] != 0), i
