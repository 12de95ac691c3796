"""The N>1 paths that have a real exchange step, run as two processes on the one GPU of the test box (gloo backend
on CUDA tensors: NCCL needs one GPU per rank): the neighbour term of update_temporal across a T-shard boundary and
the all-reduce of the spatial update, both against the single-process result."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dnmf_amd import sharding
from dnmf_amd.Demix import dNMF as M
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.manual_seed(5)
rng = np.random.RandomState(5)
sz, K, T, bs = [24, 20, 2], 6, 12, 4
pos = torch.from_numpy(rng.rand(K, 3) * np.array(sz)).float()
frames = torch.rand(T, sz[0] * sz[1] * sz[2]).cuda()
C0 = torch.rand(K, T)
jitter = torch.randn(10, 3, T) * torch.tensor([0.3, 3e-3, 3e-3, 3e-3, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4, 1e-4])[:, None, None]

def run(t0, t1, group):
    dn = M.DeformableNMF(torch.tensor(sz), K, t1 - t0, positions=pos)
    dn.verbose, dn.group = False, group
    dn.C = C0[:, t0:t1].cuda().contiguous()
    with torch.no_grad():
        dn.fp.beta += jitter[:, :, t0:t1].cuda()
    test = M.ResidentLoader(frames[t0:t1], sz, bs)
    dn.update_footprints(test, bs, sz, gamma_c=1e-2, iter_c=9, return_dense=False)     # neighbour term on
    A = dn.spatial_step(frames[t0:t1], D=torch.rand(*sz, K, generator=torch.Generator().manual_seed(1)), gamma=0.3)
    return dn.C.clone(), A.clone()

t0, t1 = sharding.shard_bounds(T, world, rank)
C_s, A_s = run(t0, t1, dist.group.WORLD)
C_f, A_f = run(0, T, None)
assert torch.allclose(C_s, C_f[:, t0:t1], rtol=1e-6, atol=0), float((C_s - C_f[:, t0:t1]).abs().max())
assert torch.allclose(A_s, A_f, rtol=2e-5, atol=1e-30), float(((A_s - A_f).abs() / (A_f.abs() + 1e-30)).max())
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


NCCL_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from dnmf_amd import sharding
from dnmf_amd.Demix import dNMF as M
rank = int(os.environ["RANK"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
world = dist.get_world_size()
torch.manual_seed(5)
rng = np.random.RandomState(5)
sz, K, T, bs = [24, 20, 2], 6, 12, 4
pos = torch.from_numpy(rng.rand(K, 3) * np.array(sz)).float()
frames = torch.rand(T, sz[0] * sz[1] * sz[2]).cuda()
C0 = torch.rand(K, T)

def run(t0, t1, group, collective="torch"):
    dn = M.DeformableNMF(torch.tensor(sz), K, t1 - t0, positions=pos)
    dn.verbose, dn.group, dn.collective = False, group, collective
    dn.C = C0[:, t0:t1].cuda().contiguous()
    A = dn.spatial_step(frames[t0:t1], D=torch.rand(*sz, K, generator=torch.Generator().manual_seed(1)), gamma=0.3)
    return A.clone(), dn

t0, t1 = sharding.shard_bounds(T, world, rank)
A_f, _ = run(0, T, None)
A_t, dn_t = run(t0, t1, dist.group.WORLD)      # ONE torch.distributed all-reduce (RCCL) of the packed A1 | C_s buffer
assert dn_t._comm is None
assert torch.allclose(A_t, A_f, rtol=2e-5, atol=1e-30), float(((A_t - A_f).abs() / (A_f.abs() + 1e-30)).max())
A_s, dn = run(t0, t1, dist.group.WORLD, "c1")  # ONE dnmf_allreduce_sum_f32 on the library's own RCCL communicator
assert dn._comm is not None and dn._comm.nranks == world
assert torch.allclose(A_s, A_f, rtol=2e-5, atol=1e-30), float(((A_s - A_f).abs() / (A_f.abs() + 1e-30)).max())
dn._comm.close()
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_ranks_rccl(tmp_path):
    """The same spatial update over RCCL, one GPU per rank -- through the caller's process group (the default) and
    through the library's own communicator (C1): runs where two GPUs are visible, skips on the one-GPU test box."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = tmp_path / "worker_nccl.py"
    script.write_text(NCCL_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29521", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2


def test_two_ranks_on_one_gpu(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29519", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("ok") == 2
