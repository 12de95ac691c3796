#!/usr/bin/env python3
"""The fit loop of the reference's demo (simulate -> loaders -> alternate motion / footprint steps) without the
plots, against this repository's drop-in ``Demix.dNMF``.  Prints how well the recovered traces match the ground
truth.  Needs an MI355X.

    python examples/demo_headless.py [--outer 5] [--epochs 10] [--iter-c 50]
"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.optim as optim
from torch.utils.data import DataLoader

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from Demix.dNMF import DeformableNMF, ExponentialFP, SimulatedVideoDataset  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--outer", type=int, default=5)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--iter-c", type=int, default=50)
    ap.add_argument("--quiet", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    np.random.seed(0)
    K, T, sz = 10, 100, torch.tensor([50, 50, 2])
    efp = ExponentialFP(sz, K, T, positions=None, shape_std=3)
    A_tC, A_t, grid, reg = efp(np.arange(3), torch.rand(K, T))
    dataset = SimulatedVideoDataset(K=K, T=T, sz=sz, shape_std=3, density=.2, bg_snr=-120, motion='gp', traces='exp',
                                    motion_par={'sigma': [5, 5, .01], 'ls': [10, 10, 10]})
    batch_size = 4
    dataloader = DataLoader(dataset, batch_size=batch_size, shuffle=True, num_workers=0)
    testloader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=0)
    dnmf = DeformableNMF(sz, K, T, positions=dataset.positions[:, :, 0])
    dnmf.verbose = not a.quiet
    optimizer = optim.Adam([dnmf.fp.beta], lr=1e-5)
    for _ in range(a.outer):
        dnmf.update_motion(dataloader, optimizer, gamma=1, epochs=a.epochs)
        A_t, Y_i, Y = dnmf.update_footprints(testloader, batch_size, sz, gamma_c=0, iter_c=a.iter_c)
    C = dnmf.C.cpu().numpy()
    corr = [np.corrcoef(C[k], dataset.traces[k])[0, 1] for k in range(K)]
    print("A_t", A_t.shape, "Y_i", Y_i.shape, "Y", Y.shape)
    print("trace correlation with ground truth: min %.3f  median %.3f" % (min(corr), float(np.median(corr))))
    return corr


if __name__ == "__main__":
    main()
