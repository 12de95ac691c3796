#!/usr/bin/env python3
"""The position initialiser end to end: simulate a moving video, register it piecewise-rigidly (``MotionCorrect``, the
class of the reference's ``Demix/MotionCorrect.py``, on the GPU), move the first frame's neuron centres by the shifts of
their patches (``apply_shifts_points``) and compare with the simulator's own per-frame centres.  Needs an MI355X.

    python examples/init_positions.py [--size 128] [--neurons 30] [--frames 50] [--stride 12]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from Demix.dNMF import DeformableNMF, SimulatedVideoDataset  # noqa: E402
from Demix.MotionCorrect import MotionCorrect  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--neurons", type=int, default=30)
    ap.add_argument("--frames", type=int, default=50)
    ap.add_argument("--stride", type=int, default=12, help="patch stride in x and y (patches of 1.5 strides; default 12: the simulator's motion varies over ~10 voxels)")
    a = ap.parse_args()
    torch.manual_seed(0)
    np.random.seed(0)
    K, T, sz = a.neurons, a.frames, torch.tensor([a.size, a.size, 2])
    dataset = SimulatedVideoDataset(K=K, T=T, sz=sz, shape_std=3, density=.2, bg_snr=-120, motion='gp', traces='exp',
                                    motion_par={'sigma': [5, 5, .01], 'ls': [10, 10, 10]})
    video = np.moveaxis(np.asarray(dataset.video), -1, 0)            # (X, Y, Z, T) -> (T, X, Y, Z)
    truth = np.asarray(dataset.positions)                            # (K, 3, T)
    stride = a.stride
    mc = MotionCorrect(video, max_shifts=(12, 12, 1), strides=(stride, stride, 1), overlaps=(stride // 2, stride // 2, 1),
                       max_deviation_rigid=3, is3D=True, pw_rigid=True)
    mc.motion_correct()                                              # template=None: rigid pass first
    P_T = mc.apply_shifts_points(video, truth[:, :, 0])              # (K, 3, T)
    still = np.abs(truth[:, :2, :] - truth[:, :2, :1]).mean()        # error of "the neurons do not move"
    err = np.abs(P_T[:, :2, :] - truth[:, :2, :]).mean()
    print(f"video {tuple(video.shape)}, {len(mc.x_shifts_els[0])} patches; rigid shifts up to "
          f"{np.abs(np.array(mc.shifts_rig)).max():.1f} voxels")
    print(f"mean |x, y error| of the per-frame centres: {err:.2f} voxels with the initialiser, {still:.2f} without")
    dn = DeformableNMF(sz, K, T, positions=torch.from_numpy(P_T[:, :, 0]).float())
    print("DeformableNMF accepts the positions:", tuple(dn.fp.A.shape))
    return err, still


if __name__ == "__main__":
    main()
