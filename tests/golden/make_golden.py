"""Generate the golden fixtures ``tests/golden/G*.npz`` by running the reference on CPU.

Run once in the build container (the GPU box has no /root/reference):

    python tests/golden/make_golden.py

The reference (``/root/reference``, read-only) is imported as it is.  Two accommodations, neither
touching its files: ``Demix/dNMF.py:7`` imports ``Methods.Demix.WUtils`` -- a package path that does
not exist in the tree -- so ``WUtils`` is registered under that name in ``sys.modules``; and the module
global ``device`` (``Demix/dNMF.py:16``, 'cuda') is set to 'cpu' after import (it is read at call time).

Only data (inputs and the reference's outputs) is written; no reference source is copied.
Fixture labels follow SURVEY.md section 8(c): G1..G10.
"""
import os
import sys
import types
import warnings

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    sys.path.insert(0, REF)
    import WUtils  # noqa: E402
    from WUtils import Simulator  # noqa: F401,E402
    for name in ("Methods", "Methods.Demix"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["Methods.Demix.WUtils"] = WUtils
    import Demix.dNMF as R  # noqa: E402
    R.device = "cpu"
    return R, Simulator


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def test_betas(T):
    """identity / affine (1.5 px shift, +-1 % linear) / small quadratic / large shift (sources leave
    the volume) / mixed -- cycled over T frames."""
    b = torch.cat((torch.zeros(1, 3), torch.eye(3), torch.zeros(6, 3)), 0)[:, :, None].repeat(1, 1, T)
    for t in range(T):
        m = t % 5
        if m == 1:
            b[0, :, t] = torch.tensor([1.5, -0.75, 0.1])
            b[1, 0, t], b[2, 1, t] = 1.01, 0.99
            b[2, 0, t] = 0.01
        elif m == 2:
            b[0, :, t] = torch.tensor([0.3, 0.2, 0.05])
            b[4, 0, t], b[5, 1, t], b[7, 0, t], b[7, 1, t] = 2e-3, -1.5e-3, 1e-3, -2e-3
            b[8, 2, t], b[9, 0, t], b[6, 2, t] = 3e-3, 2e-3, 1e-2
        elif m == 3:
            b[0, :, t] = torch.tensor([-4.25, 3.5, 0.6])
        elif m == 4:
            b[0, :, t] = torch.tensor([0.5, 0.5, 0.0])
            b[1, 1, t], b[2, 0, t] = 0.02, -0.02
    return b


def main():
    warnings.filterwarnings("ignore")
    R, Sim = load_reference()
    F = torch.nn.functional

    # ---- G1: constructor ------------------------------------------------------------------------
    sz = torch.tensor([12, 10, 2])
    pos = torch.tensor([[3.0, 4.0, 0.5], [8.2, 2.1, 1.0], [6.0, 7.5, 0.0]])
    fp = R.ExponentialFP(sz, 3, 5, positions=pos)
    save("G1_init", sz=sz.numpy(), positions=pos.numpy(), shape_std=3.0, A=fp.A.numpy(), lattice=fp.flow_id.numpy(),
         basis=fp.transformed.numpy(), beta=fp.beta.detach().numpy(), sigma=fp.sigma.numpy(),
         basis_probe=R.ExponentialFP.quadratic_basis(torch.tensor([[[[3.0, 4.0, 1.0]]]])).numpy())

    # ---- G2: forward, G7: log_det_jac ------------------------------------------------------------
    torch.manual_seed(1)
    T = 5
    C = torch.rand(3, T)
    fp = R.ExponentialFP(sz, 3, T, positions=pos)
    with torch.no_grad():
        fp.beta.copy_(test_betas(T))
    times = [0, 1, 2, 3, 4]
    A_tC, A_t, grid, reg = fp(times, C)
    ldj = np.array([[float(R.ExponentialFP.log_det_jac(fp.beta[:, :, t], fp.sz - 1)),
                     float(R.ExponentialFP.log_det_jac(fp.beta[:, :, t], fp.sz * 0))] for t in times])
    save("G2_forward", sz=sz.numpy(), positions=pos.numpy(), beta=fp.beta.detach().numpy(), C=C.numpy(),
         times=np.array(times), A_tC=A_tC.detach().numpy(), A_t=A_t.detach().numpy(), grid=grid.detach().numpy(),
         reg=reg.numpy(), log_det_jac=ldj)

    # ---- G8: simulator ---------------------------------------------------------------------------
    torch.manual_seed(0)
    np.random.seed(0)
    szv = [16, 16, 2]
    par = dict(K=3, T=6, sz=szv, shape_std=3, density=.2, bg_snr=-120, traces="exp", motion="gp",
               motion_par={"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    video, positions, traces = Sim.generate_video(**par)
    np.random.seed(3)
    tr_alone = Sim.simulate_exponential_traces(4, 30, .2)
    cell = Sim.simulate_cell(np.array([16, 16, 2, 1]), np.array([5.3, 9.1, 0.4]), 3 * np.eye(3), np.array([1.7]),
                             np.array([0]), np.array([0]), 0)
    # the same video at a noise level that is visible (bg_snr=-20) pins the noise path too
    torch.manual_seed(0)
    np.random.seed(0)
    par2 = dict(par, bg_snr=-20)
    video_n, _, _ = Sim.generate_video(**par2)
    save("G8_simulator", sz=np.array(szv), video=video.numpy(), positions=positions.numpy(), traces=traces,
         traces_alone=tr_alone, cell=cell, cell_mean=np.array([5.3, 9.1, 0.4]), cell_amp=1.7, video_noisy=video_n.numpy())

    # ---- shared small problem for G3/G4/G6/G9 -----------------------------------------------------
    torch.manual_seed(0)
    np.random.seed(0)
    szp = torch.tensor([20, 16, 2])
    K, T = 4, 8
    ds = R.SimulatedVideoDataset(K=K, T=T, sz=szp, shape_std=3, density=.2, bg_snr=-120, motion="gp", traces="exp",
                                 motion_par={"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    video = ds.video.clone()
    video[video < 0] = 0  # what __getitem__ does in place to every frame it serves (dNMF.py:214-215)
    pos0 = ds.positions[:, :, 0]

    # ---- G3: beta.grad ---------------------------------------------------------------------------
    torch.manual_seed(2)
    Cg = torch.rand(K, T)
    g3 = dict(sz=szp.numpy(), positions=pos0.numpy(), video=video.numpy(), C=Cg.numpy())
    for label, beta0, times in (("id_b1", None, [2]), ("id_b3", None, [0, 3, 5]),
                                ("pert_b1", test_betas(T), [1]), ("pert_b3", test_betas(T), [1, 2, 4]),
                                ("pert_b4", test_betas(T), [3, 4, 6, 7])):
        fp = R.ExponentialFP(szp, K, T, positions=pos0)
        if beta0 is not None:
            with torch.no_grad():
                fp.beta.copy_(beta0)
        A_tC, _, _, _ = fp(times, Cg)
        frames = video[:, :, :, times].permute(3, 0, 1, 2)
        loss = F.mse_loss(A_tC, frames)
        loss.backward()
        g3[label + "_beta"] = fp.beta.detach().numpy()
        g3[label + "_times"] = np.array(times)
        g3[label + "_grad"] = fp.beta.grad.numpy()
        g3[label + "_loss"] = float(loss)
    save("G3_grad", **g3)

    # ---- G6: spatial_pushforward, G4: update_temporal ---------------------------------------------
    dn = R.DeformableNMF(szp, K, T, positions=pos0)
    with torch.no_grad():
        dn.fp.beta.copy_(test_betas(T))
    dn.C = Cg.clone()
    loader = torch.utils.data.DataLoader(ds, batch_size=3, shuffle=False, num_workers=0)
    with torch.no_grad():
        A_t, Y_i, Y = R.ExponentialFP.spatial_pushforward(loader, 3, szp, "cpu", dn)
    # the reference sizes these by len(loader)*batch_size = 9 > T (and update_footprints then raises a
    # broadcast error, dNMF.py:143, whenever batch_size does not divide T); keep the T real frames
    save("G6_pushforward", sz=szp.numpy(), positions=pos0.numpy(), beta=dn.fp.beta.detach().numpy(), C=Cg.numpy(),
         video=video.numpy(), A_t=A_t[..., :T].astype(np.float32), Y=Y[..., :T].astype(np.float32),
         Y_i=Y_i[..., :T].astype(np.float32), padded_T=A_t.shape[-1], D=dn.D)
    A_t, Y = A_t[..., :T], Y[..., :T]
    g4 = dict(C0=Cg.numpy().astype(np.float64))
    for label, gamma in (("none", None), ("zero", 0), ("g1e2", 1e-2)):
        Cc = Cg.numpy().astype(np.float64)
        for it in range(50):
            Cc = R.DeformableNMF.update_temporal(A_t, Cc, Y, gamma=gamma)
            if it == 0:
                g4[label + "_it1"] = Cc.copy()
        g4[label + "_it50"] = Cc
    save("G4_temporal", **g4)

    # ---- G5: update_spatial ----------------------------------------------------------------------
    rng = np.random.RandomState(5)
    m, n, k, t = 9, 7, 3, 11
    A5, C5, Y5, D5 = rng.rand(m, n, k), rng.rand(k, t), rng.rand(m, n, t), rng.rand(m, n, k)
    save("G5_spatial", A=A5, C=C5, Y_i=Y5, D=D5, out_noD=R.DeformableNMF.update_spatial(A5, C5, Y5),
         out_D=R.DeformableNMF.update_spatial(A5, C5, Y5, D=D5, gamma=0.7))

    # ---- G9: outer iterations of the demo loop -----------------------------------------------------
    g9 = dict(sz=szp.numpy(), positions=pos0.numpy(), video=video.numpy())
    for label, lr, epochs, shuffle, bs in (("lr1e-5_ordered", 1e-5, 1, False, 4), ("lr1e-3_ordered", 1e-3, 2, False, 4),
                                           ("lr1e-3_shuffled", 1e-3, 3, True, 2)):
        torch.manual_seed(7)
        dn = R.DeformableNMF(szp, K, T, positions=pos0)
        C0 = dn.C.clone()
        opt = torch.optim.Adam([dn.fp.beta], lr=lr)
        gen = torch.Generator().manual_seed(11)
        train = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=0, generator=gen)
        test = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=False, num_workers=0)
        order = []
        if shuffle:  # record the batch order the seeded sampler will produce
            gen2 = torch.Generator().manual_seed(11)
            probe = torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0, generator=gen2)
            for _ in range(epochs):
                order.append([d[1].tolist() for d in probe])
        else:
            order = [[d[1].tolist() for d in test]] * epochs
        _stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")
        try:
            dn.update_motion(train, opt, gamma=1, epochs=epochs)
            beta1 = dn.fp.beta.detach().numpy().copy()
            dn.update_footprints(test, bs, szp, gamma_c=0, iter_c=5)
            dn.update_motion(train, opt, gamma=1, epochs=1) if not shuffle else None
        finally:
            sys.stdout = _stdout
        g9[label + "_C0"] = C0.numpy()
        g9[label + "_beta_after_motion"] = beta1
        g9[label + "_C_after_footprints"] = dn.C.numpy()
        g9[label + "_beta_after_second_motion"] = dn.fp.beta.detach().numpy()
        g9[label + "_order"] = np.array([[b + [-1] * (bs - len(b)) for b in ep] for ep in order])
        g9[label + "_cfg"] = np.array([lr, epochs, float(shuffle), bs])
    save("G9_loop", **g9)

    # ---- G10: duplicated-slice Z=2 run that defines Z=1 --------------------------------------------
    sz2 = torch.tensor([14, 11, 2])
    K, T = 3, 4
    torch.manual_seed(4)
    pos2 = torch.tensor([[3.0, 4.0, 0.0], [9.5, 2.5, 0.0], [6.0, 8.0, 0.0]])
    fp = R.ExponentialFP(sz2, K, T, positions=pos2)
    A2d = fp.A[:, :, 0, :].clone()
    fp.A = torch.stack((A2d, A2d), 2)  # both slices identical (the constructor's Gaussian has a z term)
    b = test_betas(T)
    b[:, 2, :] = 0
    b[3, 2, :] = 1          # z row/column at identity
    b[3, :2, :] = 0
    b[6, :, :] = 0
    b[8:, :, :] = 0          # no z-dependent terms in x,y
    with torch.no_grad():
        fp.beta.copy_(b)
    C = torch.rand(K, T)
    Y2d = torch.rand(14, 11, T)
    Yv = torch.stack((Y2d, Y2d), 2)
    times = list(range(T))
    A_tC, A_t, grid, _ = fp(times, C)
    loss = F.mse_loss(A_tC, Yv.permute(3, 0, 1, 2))
    loss.backward()
    A_t64 = np.transpose(A_t.detach().numpy().astype(np.float64), [2, 3, 4, 1, 0])
    Cc = C.numpy().astype(np.float64)
    for _ in range(5):
        Cc = R.DeformableNMF.update_temporal(A_t64, Cc, Yv.numpy().astype(np.float64), gamma=0)
    save("G10_2d", sz=sz2.numpy(), A2d=A2d.numpy(), beta=b.numpy(), C=C.numpy(), Y2d=Y2d.numpy(),
         A_t=A_t.detach().numpy(), A_tC=A_tC.detach().numpy(), loss=float(loss), grad=fp.beta.grad.numpy(),
         C_it5=Cc)


if __name__ == "__main__":
    main()
