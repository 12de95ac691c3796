"""Host-side logic for sharding the T axis over ranks (one process per GPU).

Frames are independent in both fit steps, so a rank owns a contiguous block of frames together with its columns of
beta, of the Adam moments and of C; the footprints are replicated.  The only thing ranks must agree on is the
GLOBAL mini-batch sequence of the motion step: every ``optimizer.step()`` of the reference moves every column of
beta, so the number of steps a column sees before and after its own mini-batch depends on the global batch order
(hard part 4).  All ranks therefore draw the same permutation (same seed) and each keeps the
entries that fall into its block.  Nothing here touches the GPU.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


def shard_bounds(T_total: int, world: int, rank: int):
    """Contiguous block [t0, t1) of rank ``rank``; block sizes differ by at most one frame."""
    base, extra = divmod(int(T_total), int(world))
    t0 = rank * base + min(rank, extra)
    return t0, t0 + base + (1 if rank < extra else 0)


@dataclass
class EpochPlan:
    """One epoch of mini-batches as seen by one shard.

    nsteps       optimiser steps in the epoch (global number of mini-batches)
    frame_step   (T_local,) int32: 0-based global index of the mini-batch that holds each local frame, -1 if none
    order        (T_local,) int32: the local frames sorted by frame_step (threads of the Adam kernel that run together
                 then walk through the same window of steps)
    groups       list of (local frame indices (int64, epoch order), frames per mini-batch): frames whose
                 mini-batches have the same size share one kernel launch (the mean of the loss runs over that size)
    batches      per global mini-batch: local frame indices (possibly empty) -- the step-by-step view
    """
    nsteps: int
    frame_step: torch.Tensor
    order: torch.Tensor
    groups: list
    batches: list


def plan_epoch(perm: torch.Tensor, batch_size: int, t0: int, t1: int) -> EpochPlan:
    """Split the global frame order ``perm`` (1-D int64, a permutation of 0..T_total-1 or any visiting order)
    into mini-batches of ``batch_size`` and keep what belongs to the block [t0, t1).

    numpy on purpose: these are a few thousand elements, and torch's CPU kernels would wake its whole intra-op
    thread pool once per epoch -- on a host whose CPU quota is smaller than its core count that gets the process
    throttled for most of a scheduler period, longer than the epoch itself takes on the GPU."""
    perm = perm.to(torch.int64).cpu().numpy()
    n_total = perm.size
    nsteps = (n_total + batch_size - 1) // batch_size
    step_of_pos = np.arange(n_total, dtype=np.int64) // batch_size
    tail = n_total % batch_size
    mine = (perm >= t0) & (perm < t1)
    local = perm[mine] - t0
    steps = step_of_pos[mine]            # non-decreasing: positions are visited in order
    frame_step = np.full((t1 - t0,), -1, dtype=np.int32)
    frame_step[local] = steps.astype(np.int32)
    sizes = np.where(steps == nsteps - 1, tail if tail else batch_size, batch_size)
    groups = [(torch.from_numpy(local[sizes == s]), int(s)) for s in sorted({batch_size, tail or batch_size}, reverse=True)]
    groups = [g for g in groups if g[0].numel()]
    counts = np.bincount(steps, minlength=nsteps).tolist()
    order = np.argsort(frame_step, kind="stable").astype(np.int32)
    return EpochPlan(nsteps=nsteps, frame_step=torch.from_numpy(frame_step), order=torch.from_numpy(order), groups=groups,
                     batches=_LazySplit(torch.from_numpy(local), counts))


def plan_from_batches(batches, T_local: int):
    """The EpochPlan of an epoch whose mini-batches are given explicitly (what a stock DataLoader served): ``batches``
    is a list of lists of LOCAL frame indices, one per optimiser step, of any sizes.  Returns None when a frame occurs
    more than once in the epoch (the per-column form of the Adam epoch then does not apply)."""
    nsteps = len(batches)
    flat = np.fromiter((i for b in batches for i in b), dtype=np.int64)
    if flat.size == 0 or np.unique(flat).size != flat.size or flat.min() < 0 or flat.max() >= T_local:
        return None
    sizes_per_batch = np.fromiter((len(b) for b in batches), dtype=np.int64, count=nsteps)
    steps = np.repeat(np.arange(nsteps, dtype=np.int64), sizes_per_batch)
    frame_step = np.full((T_local,), -1, dtype=np.int32)
    frame_step[flat] = steps.astype(np.int32)
    sizes = sizes_per_batch[steps]
    groups = [(torch.from_numpy(flat[sizes == s]), int(s)) for s in sorted(set(sizes_per_batch.tolist()), reverse=True) if s > 0]
    order = np.argsort(frame_step, kind="stable").astype(np.int32)
    return EpochPlan(nsteps=nsteps, frame_step=torch.from_numpy(frame_step), order=torch.from_numpy(order), groups=groups,
                     batches=_LazySplit(torch.from_numpy(flat), sizes_per_batch.tolist()))


class _LazySplit:
    """``list(torch.split(local, counts))`` built on first use: the fused epoch never looks at it."""

    def __init__(self, local, counts):
        self._local, self._counts, self._parts = local, counts, None

    def _get(self):
        if self._parts is None:
            self._parts = list(torch.split(self._local, self._counts))
        return self._parts

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._counts)

    def __getitem__(self, i):
        return self._get()[i]
