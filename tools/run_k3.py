#!/usr/bin/env python3
"""Drive only the K3 Gram kernel at the bench geometry (for rocprofv3 counter passes).

    python tools/run_k3.py [--frames 1000] [--reps 3] [--size 512] [--neurons 100] [--z 1]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--neurons", type=int, default=100)
    ap.add_argument("--z", type=int, default=1)
    ap.add_argument("--jitter", type=float, default=1e-3, help="random perturbation of beta (0 = identity warp)")
    ap.add_argument("--sparse", action="store_true", help="run the zero-skipping kernel K3s instead of K3")
    ap.add_argument("--bf16", action="store_true", help="run K3b (bf16 operands) instead of K3")
    ap.add_argument("--lists", action="store_true", help="run the neuron-list kernel K3n instead of K3")
    a = ap.parse_args()
    from dnmf_amd import ops
    from dnmf_amd.Demix import dNMF as M
    torch.manual_seed(0)
    sz = [a.size, a.size, a.z]
    K, T = a.neurons, a.frames
    pos = torch.rand(K, 3) * torch.tensor([float(a.size), float(a.size), float(a.z - 1)])
    fp = M.ExponentialFP(torch.tensor(sz), K, T, positions=pos)
    with torch.no_grad():
        scale = torch.tensor([1.0, 1e-3, 1e-3, 1e-3, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6, 1e-6], device="cuda")
        fp.beta += a.jitter * scale[:, None, None] * torch.randn_like(fp.beta)
    frames = torch.rand(T, fp.P, device="cuda")
    ws = None
    ev = []
    for _ in range(a.reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        if a.lists:
            ly = fp.packed_lists()
            G, r, ws = ops.warp_gram_rhs_lists(ly, K, sz, fp.beta.detach(), None, frames, workspace=ws)
        elif a.sparse:
            sp = fp.packed_sparse()
            G, r, ws = ops.warp_gram_rhs_sparse(sp["Aps"], K, sp["order"], sp["row_mask"], sz, fp.beta.detach(), None,
                                                frames, workspace=ws)
        else:
            G, r, ws = ops.warp_gram_rhs(fp.packed_footprints(), K, sz, fp.beta.detach(), None, frames, workspace=ws,
                                         bf16=a.bf16)
        e.record()
        ev.append((s, e))
    torch.cuda.synchronize()
    ms = [s.elapsed_time(e) for s, e in ev]
    P = fp.P
    ntap = 8 if a.z > 1 else 4
    flops = T * (P * K * (K + 1) + 2 * P * K + 2 * ntap * P * K)
    if a.sparse:
        print("occupancy", fp.packed_sparse()["occupancy"])
    if a.lists:
        print("nslot", fp.packed_lists()["nslot"], "boxfrac", fp.packed_lists()["boxfrac"])
    print(f"K3{'n' if a.lists else 's' if a.sparse else 'b' if a.bf16 else ''} {sz} K={K} T={T}: ms per launch {['%.2f' % m for m in ms]}  -> {flops / (min(ms) * 1e-3) / 1e12:.1f} TFLOP/s algorithmic")


if __name__ == "__main__":
    main()
