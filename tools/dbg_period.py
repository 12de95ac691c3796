import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def cg():
    out = {}
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            out[f] = open(f).read().replace("\n", " ")
        except Exception as e:
            pass
    return out
print("threads", torch.get_num_threads(), "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), cg(), flush=True)
from dnmf_amd import ops
from dnmf_amd.Demix import dNMF as M
from dnmf_amd.WUtils import Simulator
size, K, Tn, bs = 512, 100, 4000, 4
sz = [size, size, 1]
torch.manual_seed(0); np.random.seed(0)
frames, positions, _ = Simulator.generate_video_resident(K, Tn, sz, 3, .2, -120, {"sigma": [5, 5, .01], "ls": [10, 10, 10]})
frames.clamp_(min=0)
dn = M.DeformableNMF(torch.tensor(sz), K, Tn, positions=positions[:, :, 0].contiguous()); dn.verbose = False
opt = torch.optim.Adam([dn.fp.beta], lr=1e-5)
train = M.ResidentLoader(frames, sz, bs, shuffle=True, generator=torch.Generator().manual_seed(1))
test = M.ResidentLoader(frames, sz, bs)
def sweep():
    dn.update_motion(train, opt, gamma=1, epochs=1)
    dn.update_footprints(test, bs, sz, gamma_c=0, iter_c=50, return_dense=False)
sweep()
def run(label, n=12):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); sweep(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
    print(label, ["%.1f" % t for t in ts], flush=True)
run("default threads")
print(cg(), flush=True)
torch.set_num_threads(8)
run("8 threads")
print(cg(), flush=True)
torch.set_num_threads(1)
run("1 thread")
print(cg(), flush=True)
# pure busy loop without torch work
for rep in range(3):
    t0 = time.perf_counter(); gaps = []
    last = t0
    while time.perf_counter() - t0 < 0.3:
        now = time.perf_counter()
        if now - last > 0.005:
            gaps.append(round(1e3 * (now - last), 1))
        last = now
    print("busy-loop gaps > 5 ms:", gaps, flush=True)
