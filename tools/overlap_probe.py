#!/usr/bin/env python3
"""Does the store-bound reconstruction kernel overlap with the ALU-bound Gram kernel when they run on two streams?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dnmf_amd import ops
from dnmf_amd.Demix import dNMF as M
torch.manual_seed(0)
size, K, T = 512, 100, 4000
sz = [size, size, 1]
pos = torch.rand(K, 3) * torch.tensor([float(size), float(size), 0.0])
fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
with torch.no_grad():
    fp.beta += 1e-3 * torch.tensor([1.0, 1e-3, 1e-3, 1e-3, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6], device="cuda")[:, None, None] * torch.randn_like(fp.beta)
frames = torch.rand(T, fp.P, device="cuda")
C = torch.rand(K, T, device="cuda")
S = torch.empty(T, ops.halo_voxels(sz), device="cuda")
ly = fp.packed_lists()
times = torch.arange(T, dtype=torch.int32, device="cuda")
half = T // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ws = [None, None]
def gram(lo, hi, i):
    _, _, ws[i] = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), times[lo:hi], frames[lo:hi], workspace=ws[i])
wk = [None]
grad = torch.zeros_like(fp.beta)
def recon(lo, hi):
    ops.recon_image_lists(ly, K, sz, C, times[lo:hi], out=S[lo:hi])
    out = ops.warp_recon_grad(S[lo:hi], None, frames[lo:hi], None, sz, fp.beta.detach(), times[lo:hi], grad=grad, want_loss=False,
                              want_reg=False, workspace=wk[0], norm_frames=4)
    wk[0] = out["workspace"]
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
def serial():
    gram(0, half, 0); recon(half, T)
def overlapped():
    with torch.cuda.stream(s1): gram(0, half, 0)
    with torch.cuda.stream(s2): recon(half, T)
print("gram half     %.3f ms" % timeit(lambda: gram(0, half, 0)))
print("recon + K2 half %.3f ms" % timeit(lambda: recon(half, T)))
print("serial        %.3f ms" % timeit(serial))
print("two streams   %.3f ms" % timeit(overlapped))
