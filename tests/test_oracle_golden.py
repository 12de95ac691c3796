"""The CPU oracle (oracle/dnmf_oracle.py) against the fixtures captured from the reference.

Every tolerance is stated next to its assertion.  The fixtures were produced by
tests/golden/make_golden.py from the reference's own Demix/dNMF.py and WUtils/Simulator.py.
"""
import numpy as np
import pytest
import torch

from oracle import dnmf_oracle as O
from conftest import golden


def test_G1_constructor():
    g = golden("G1_init")
    sz = g["sz"]
    lat = O.voxel_lattice(sz)
    np.testing.assert_array_equal(lat, g["lattice"])
    np.testing.assert_array_equal(O.quadratic_basis(lat), g["basis"])
    np.testing.assert_array_equal(O.quadratic_basis(np.array([[[[3.0, 4.0, 1.0]]]])), g["basis_probe"])
    np.testing.assert_array_equal(O.identity_beta(5), g["beta"])
    A = O.gaussian_footprints(sz, g["positions"], g["sigma"])
    np.testing.assert_array_equal(A, g["A"])  # bit-exact: same torch-CPU op sequence


@pytest.mark.parametrize("sampler", [O.trilinear_sample, O.trilinear_sample_torch])
def test_G2_forward(sampler):
    g = golden("G2_forward")
    sz = g["sz"]
    lat = O.voxel_lattice(sz)
    A = O.gaussian_footprints(sz, g["positions"], np.full(3, 3.0))
    A_tC, A_t, n, reg = O.forward(A, O.quadratic_basis(lat), g["beta"], sz, g["times"].tolist(), g["C"], sampler)
    np.testing.assert_array_equal(n, g["grid"])          # same einsum + normalise sequence
    # the numpy restatement of the ATen sampler differs from the library only in rounding order
    np.testing.assert_allclose(A_t, g["A_t"], rtol=0, atol=2e-7)
    np.testing.assert_allclose(A_tC, g["A_tC"], rtol=0, atol=5e-7)
    np.testing.assert_allclose(reg, g["reg"], rtol=1e-5, atol=1e-9)


def test_G7_log_det_jac():
    g = golden("G2_forward")
    sz = g["sz"].astype(np.float32)
    for t in range(g["beta"].shape[2]):
        got = [O.log_det_jac(g["beta"][:, :, t], sz - 1), O.log_det_jac(g["beta"][:, :, t], sz * 0)]
        np.testing.assert_allclose(got, g["log_det_jac"][t], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("label", ["id_b1", "id_b3", "pert_b1", "pert_b3", "pert_b4"])
def test_G3_beta_grad(label):
    g = golden("G3_grad")
    sz = g["sz"]
    lat = O.voxel_lattice(sz)
    basis = O.quadratic_basis(lat)
    A = O.gaussian_footprints(sz, g["positions"], np.full(4, 3.0))
    times = g[label + "_times"].tolist()
    frames = np.moveaxis(g["video"][..., times], -1, 0)
    loss, grad = O.mse_beta_grad_autograd(A, basis, g[label + "_beta"], sz, times, g["C"], frames)
    np.testing.assert_allclose(loss, g[label + "_loss"], rtol=1e-6)
    np.testing.assert_allclose(grad, g[label + "_grad"], rtol=1e-5, atol=1e-9)
    # hand-written gradient (the formula the HIP kernel implements).  On the identity lattice it only
    # matches if the fp32 normalise/un-normalise round trip is reproduced (hard part 1)
    loss2, grad2 = O.mse_beta_grad_analytic(A, basis, g[label + "_beta"], sz, times, g["C"], frames)
    np.testing.assert_allclose(loss2, g[label + "_loss"], rtol=1e-5)
    scale = np.abs(g[label + "_grad"]).max()
    np.testing.assert_allclose(grad2, g[label + "_grad"], rtol=1e-4, atol=1e-5 * scale)


@pytest.mark.parametrize("label,gamma", [("none", None), ("zero", 0), ("g1e2", 1e-2)])
def test_G4_update_temporal(label, gamma):
    g4, g6 = golden("G4_temporal"), golden("G6_pushforward")
    A_t, Y = g6["A_t"].astype(np.float64), g6["Y"].astype(np.float64)
    C = g4["C0"].copy()
    C1 = O.update_temporal(A_t, C, Y, gamma=gamma)
    np.testing.assert_allclose(C1, g4[label + "_it1"], rtol=1e-13)
    for _ in range(49):
        C1 = O.update_temporal(A_t, C1, Y, gamma=gamma)
    np.testing.assert_allclose(C1, g4[label + "_it50"], rtol=1e-11)
    # hoisted Gram/rhs form (what the HIP path runs) is the same iteration
    G, r = O.gram_rhs(A_t, Y)
    np.testing.assert_allclose(O.mu_temporal_from_gram(G, r, C, gamma, 50), g4[label + "_it50"], rtol=1e-11)


def test_G5_update_spatial():
    g = golden("G5_spatial")
    np.testing.assert_allclose(O.update_spatial(g["A"], g["C"], g["Y_i"]), g["out_noD"], rtol=1e-13)
    np.testing.assert_allclose(O.update_spatial(g["A"], g["C"], g["Y_i"], D=g["D"], gamma=0.7), g["out_D"], rtol=1e-13)


def test_G6_pushforward():
    g = golden("G6_pushforward")
    sz = g["sz"]
    m = O.OracleModel(sz, 4, 8, g["positions"], C0=g["C"])
    with torch.no_grad():
        m.beta_param.copy_(torch.from_numpy(g["beta"]))
    A_t, Y_i, Y = m.pushforward(g["video"], 3, with_registration=True)
    np.testing.assert_array_equal(A_t.astype(np.float32), g["A_t"])
    np.testing.assert_array_equal(Y.astype(np.float32), g["Y"])
    np.testing.assert_array_equal(Y_i.astype(np.float32), g["Y_i"])
    np.testing.assert_allclose(O.distance_penalty(sz, g["positions"]), g["D"], rtol=1e-12)


def test_G8_simulator():
    g = golden("G8_simulator")
    torch.manual_seed(0)
    np.random.seed(0)
    video, positions, traces = O.generate_video(3, 6, g["sz"], 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    np.testing.assert_allclose(positions, g["positions"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(traces, g["traces"])
    # scipy's pdf evaluates exp(-(r2/(2 s) + log-normaliser)) * normaliser; ours exp(-r2/(2 s)): ulp-level
    np.testing.assert_allclose(video, g["video"], rtol=2e-6, atol=1e-9)
    torch.manual_seed(0)
    np.random.seed(0)
    video_n, _, _ = O.generate_video(3, 6, g["sz"], 3, .2, -20, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    np.testing.assert_allclose(video_n, g["video_noisy"], rtol=2e-6, atol=1e-7)
    np.random.seed(3)
    np.testing.assert_array_equal(O.simulate_exponential_traces(4, 30, .2), g["traces_alone"])
    np.testing.assert_allclose(O.render_cell(g["sz"], g["cell_mean"], 3, float(g["cell_amp"])), g["cell"][..., 0], rtol=1e-12)


@pytest.mark.parametrize("label", ["lr1e-5_ordered", "lr1e-3_ordered", "lr1e-3_shuffled"])
def test_G9_demo_loop(label):
    g = golden("G9_loop")
    lr, epochs, shuffled, bs = g[label + "_cfg"]
    epochs, bs = int(epochs), int(bs)
    m = O.OracleModel(g["sz"], 4, 8, g["positions"], C0=g[label + "_C0"])
    opt = torch.optim.Adam([m.beta_param], lr=float(lr))
    order = [[[i for i in b if i >= 0] for b in ep] for ep in g[label + "_order"].tolist()]
    for ep in order:
        m.update_motion(g["video"], ep, opt, gamma=1, epochs=1)
    np.testing.assert_allclose(m.beta, g[label + "_beta_after_motion"], rtol=1e-6, atol=1e-9)
    m.update_footprints(g["video"], bs, gamma_c=0, iter_c=5)
    np.testing.assert_allclose(m.C, g[label + "_C_after_footprints"], rtol=1e-5)
    if not shuffled:
        m.update_motion(g["video"], order[0], opt, gamma=1, epochs=1)
        np.testing.assert_allclose(m.beta, g[label + "_beta_after_second_motion"], rtol=1e-6, atol=1e-9)


def test_G10_z1_semantics():
    """A Z=1 oracle run equals one slice of the reference's duplicated-slice Z=2 run."""
    g = golden("G10_2d")
    X, Y, _ = g["sz"]
    sz1 = [int(X), int(Y), 1]
    A = g["A2d"][:, :, None, :]
    lat = O.voxel_lattice(sz1)
    basis = O.quadratic_basis(lat)
    T = g["beta"].shape[2]
    times = list(range(T))
    A_tC, A_t, _, _ = O.forward(A, basis, g["beta"], sz1, times, g["C"])
    for z in (0, 1):
        np.testing.assert_allclose(A_t[..., 0], g["A_t"][..., z], rtol=0, atol=2e-7)
        np.testing.assert_allclose(A_tC[..., 0], g["A_tC"][..., z], rtol=0, atol=5e-7)
    frames = np.moveaxis(g["Y2d"], -1, 0)[:, :, :, None]
    loss, grad = O.mse_beta_grad_autograd(A, basis, g["beta"], sz1, times, g["C"], frames)
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-6)
    # mean over B*P: the Z=2 run has twice the voxels and twice the (identical) terms -> same gradient
    scale = np.abs(g["grad"]).max()
    noz = [0, 1, 2, 4, 5, 7]  # basis terms without z; the z-terms vanish identically when z == 0
    np.testing.assert_allclose(grad[noz, :2], g["grad"][noz, :2], rtol=1e-4, atol=1e-6 * scale)
    assert np.all(grad[:, 2] == 0) and np.all(grad[[3, 6, 8, 9]] == 0)
    A_t64 = np.transpose(A_t.astype(np.float64), [2, 3, 4, 1, 0])
    Cc = g["C"].astype(np.float64)
    for _ in range(5):
        Cc = O.update_temporal(A_t64, Cc, g["Y2d"][:, :, None, :].astype(np.float64), gamma=0)
    np.testing.assert_allclose(Cc, g["C_it5"], rtol=1e-5)


def test_reference_polynomial_is_the_index_order_fma_chain():
    """The HIP kernels evaluate the warped coordinate as the chain of fused multiply-adds over the ten basis terms in
    index order (csrc/common.hpp: poly_a).  That is what torch's CPU einsum -- the reference's Demix/dNMF.py:54 --
    computes, bit for bit; pinned here so that a torch build with another accumulation order shows up as a failure of
    THIS test rather than as unexplained last-bit differences at lattice coincidences."""
    import torch
    rng = np.random.RandomState(0)
    for sz in ([20, 16, 2], [50, 50, 2], [64, 64, 1]):
        lat = O.voxel_lattice(sz)
        basis = O.quadratic_basis(lat)
        T = 4
        beta = O.identity_beta(T) + (rng.randn(10, 3, T) * np.array([0.7] + [1e-2] * 3 + [2e-4] * 6)[:, None, None]).astype(np.float32)
        beta[:, :, 0] = O.identity_beta(1)[:, :, 0] + np.float32(1e-3) * np.sign(rng.randn(10, 3)).astype(np.float32)  # one Adam step
        q = torch.einsum("mnza,abt->mnzbt", torch.from_numpy(basis), torch.from_numpy(beta)).numpy()
        B = basis.reshape(-1, 10)
        chain = np.zeros((B.shape[0], 3, T), np.float32)
        for a in range(10):   # fma emulated in float64: the product of two fp32 numbers is exact there
            chain = (B[:, a][:, None, None].astype(np.float64) * beta[a][None].astype(np.float64) + chain.astype(np.float64)
                     ).astype(np.float32)
        assert np.array_equal(q, chain.reshape(q.shape))
