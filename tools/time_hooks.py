import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops
from dnmf_amd.Demix import dNMF as M
torch.manual_seed(0)
size, K, Tn = 512, 100, 1000
sz = [size, size, 1]
pos = torch.rand(K, 3) * torch.tensor([512.0, 512.0, 0.0])
fp = M.ExponentialFP(torch.tensor(sz), K, Tn, positions=pos)
frames = torch.rand(Tn, fp.P, device="cuda")
sp = fp.packed_sparse()
def run(n=5):
    ws = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        G, r, ws = ops.warp_gram_rhs_sparse(sp["Aps"], K, sp["order"], sp["row_mask"], sz, fp.beta.detach(), None, frames, workspace=ws)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
run(2)
print("plain            %.2f ms" % run())
ops.TIMING = {}
print("TIMING+counters  %.2f ms" % run())
ops.SPARSE_COUNTERS = None
import dnmf_amd.ops as o
# events only: fake by keeping TIMING but stubbing counters
orig = torch.zeros
ops.TIMING = None
# counters only
lib = o._lib.load()
cnt = torch.zeros(2, dtype=torch.int64, device="cuda")
ws = torch.empty(lib.dnmf_warp_gram_rhs_sparse_workspace(fp.P, K, Tn) // 4 + 1, device="cuda")
G = torch.empty(Tn, K, K, device="cuda"); r = torch.empty(Tn, K, device="cuda")
def raw(counters, n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        lib.dnmf_warp_gram_rhs_sparse(sp["Aps"].data_ptr(), sp["Aps"].shape[1], K, sp["order"].data_ptr(), sp["row_mask"].data_ptr(), 512, 512, 1,
            fp.beta.data_ptr(), Tn, None, Tn, frames.data_ptr(), frames.stride(0), None, G.data_ptr(), r.data_ptr(), ws.data_ptr(), ws.numel() * 4,
            counters, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("raw no counters  %.2f ms" % raw(None))
print("raw counters     %.2f ms" % raw(cnt.data_ptr()))
def ev(n=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        a = torch.cuda.Event(enable_timing=True); a.record()
        lib.dnmf_warp_gram_rhs_sparse(sp["Aps"].data_ptr(), sp["Aps"].shape[1], K, sp["order"].data_ptr(), sp["row_mask"].data_ptr(), 512, 512, 1,
            fp.beta.data_ptr(), Tn, None, Tn, frames.data_ptr(), frames.stride(0), None, G.data_ptr(), r.data_ptr(), ws.data_ptr(), ws.numel() * 4,
            None, torch.cuda.current_stream().cuda_stream)
        b = torch.cuda.Event(enable_timing=True); b.record()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
print("raw events       %.2f ms" % ev())
