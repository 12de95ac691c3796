"""T-axis sharding: host logic on one process, and the N>1 paths under torch.distributed (gloo, world_size 2, CPU)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT
from dnmf_amd import sharding


def test_shard_bounds_partition_the_axis():
    for T, W in ((16000, 8), (4000, 1), (10, 3), (7, 8)):
        b = [sharding.shard_bounds(T, W, r) for r in range(W)]
        assert b[0][0] == 0 and b[-1][1] == T
        assert all(b[i][1] == b[i + 1][0] for i in range(W - 1))
        assert max(t1 - t0 for t0, t1 in b) - min(t1 - t0 for t0, t1 in b) <= 1


@pytest.mark.parametrize("T,bs,W", [(16, 4, 2), (14, 4, 3), (9, 2, 2)])
def test_epoch_plans_of_all_shards_tile_the_global_plan(T, bs, W):
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(0))
    whole = sharding.plan_epoch(perm, bs, 0, T)
    assert whole.nsteps == (T + bs - 1) // bs
    for j, b in enumerate(whole.batches):
        assert b.tolist() == perm[j * bs:(j + 1) * bs].tolist()
        assert all(int(whole.frame_step[t]) == j for t in b.tolist())
    seen = torch.full((T,), -1, dtype=torch.int32)
    for r in range(W):
        t0, t1 = sharding.shard_bounds(T, W, r)
        p = sharding.plan_epoch(perm, bs, t0, t1)
        assert p.nsteps == whole.nsteps
        seen[t0:t1] = p.frame_step
        for j, b in enumerate(p.batches):      # local members of global batch j, in the global visiting order
            assert (b + t0).tolist() == [t for t in whole.batches[j].tolist() if t0 <= t < t1]
        got = sorted((int(i) + t0, nf) for idx, nf in p.groups for i in idx.tolist())
        want = sorted((t, len(whole.batches[int(whole.frame_step[t])])) for t in range(t0, t1))
        assert got == want
    assert torch.equal(seen, whole.frame_step)


WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dnmf_amd import sharding
from dnmf_amd.WUtils import Simulator as S
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
K, T, sz = 3, 9, [12, 10, 2]
par = {"sigma": [5, 5, .01], "ls": [10, 10, 10]}
t0, t1 = sharding.shard_bounds(T, world, rank)
torch.manual_seed(0); np.random.seed(0)
noise = 1e-2 * torch.randn(T, sz[0] * sz[1] * sz[2])
np.random.seed(0)
part, pos, tr = S.generate_video_resident(K, T, sz, 3, .2, -20, par, device="cpu", t0=t0, t1=t1, noise=noise,
                                          group=dist.group.WORLD)
np.random.seed(0)
full, pos_f, tr_f = S.generate_video_resident(K, T, sz, 3, .2, -20, par, device="cpu", noise=noise)
assert torch.equal(pos, pos_f) and np.array_equal(tr, tr_f)
assert torch.allclose(part, full[t0:t1], rtol=1e-6, atol=0), float((part - full[t0:t1]).abs().max())
# every rank draws the same global permutation; plans agree on the number of steps and tile the frames
gen = torch.Generator().manual_seed(7)
perm = torch.randperm(T, generator=gen)
plan = sharding.plan_epoch(perm, 4, t0, t1)
steps = [None] * world
dist.all_gather_object(steps, (plan.nsteps, plan.frame_step.tolist()))
assert len({s[0] for s in steps}) == 1
assert sum(len(s[1]) for s in steps) == T
whole = sharding.plan_epoch(perm, 4, 0, T)
assert sum((s[1] for s in steps), []) == whole.frame_step.tolist()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_shards_reproduce_the_single_process_video(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2
