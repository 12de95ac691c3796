#!/usr/bin/env python3
"""Which fp32 operation sequence does torch.optim.Adam run on this GPU build (foreach path), bit for bit?

    python tools/adam_probe.py

One optimiser step on random (p, m, v, g) -- and one with g = 0, a "coasting" step -- is compared with candidate
sequences evaluated with fused multiply-adds emulated in float64 (the product of two fp32 numbers is exact there).
dnmf_adam_epoch's literal step (csrc/adam_epoch.hip: adam_one) encodes the sequence that matches 100 %.
"""
import itertools
import math

import torch


def f32(x):
    return x.to(torch.float32)


def fma(a, b, c):
    return f32(a.double() * b.double() + c.double())


def mul(a, b):
    return f32(a.double() * b.double())


def add(a, b):
    return f32(a.double() + b.double())


def div(a, b):
    return a / b   # IEEE fp32 division on the device


def main():
    torch.manual_seed(0)
    n = 1 << 20
    dev = "cuda"
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    for label, zero_grad in (("gradient step", False), ("zero-gradient step", True)):
        p0 = torch.randn(n, device=dev)
        m0 = torch.randn(n, device=dev) * 1e-3
        v0 = torch.rand(n, device=dev) * 1e-6
        g = torch.zeros(n, device=dev) if zero_grad else torch.randn(n, device=dev) * 1e-3
        step = 7
        p = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([p], lr=lr, betas=(b1, b2), eps=eps)
        opt.state[p] = {"step": torch.tensor(float(step - 1)), "exp_avg": m0.clone(), "exp_avg_sq": v0.clone()}
        p.grad = g.clone()
        opt.step()
        st = opt.state[p]
        m_t, v_t, p_t = st["exp_avg"], st["exp_avg_sq"], p.detach()
        w = torch.tensor(1.0 - b1, dtype=torch.float32, device=dev)       # what lerp_ receives as a Python float
        omb2 = torch.tensor(1.0 - b2, dtype=torch.float32, device=dev)
        b2f = torch.tensor(b2, dtype=torch.float32, device=dev)
        bc1 = 1.0 - b1 ** step
        bc2s = math.sqrt(1.0 - b2 ** step)
        step_size = torch.tensor(-(lr / bc1), dtype=torch.float32, device=dev)   # foreach: step_size = (lr / bc1).neg()
        bc2t = torch.tensor(bc2s, dtype=torch.float32, device=dev)
        epst = torch.tensor(eps, dtype=torch.float32, device=dev)
        d = add(g, -m0)
        cands_m = {"m + w*(g-m)": add(m0, mul(w, d)), "fma(w, g-m, m)": fma(w, d, m0)}
        for k, val in cands_m.items():
            print(f"{label:20s} m: {k:28s} equal {float((val == m_t).float().mean()):.6f}")
        vb = mul(v0, b2f)
        gg = mul(g, g)
        cands_v = {"b2 v + omb2*(g g)": add(vb, mul(omb2, gg)), "fma(omb2, g g, b2 v)": fma(omb2, gg, vb),
                   "fma(omb2 g, g, b2 v)": fma(mul(omb2, g), g, vb), "b2 v + (omb2 g) g": add(vb, mul(mul(omb2, g), g))}
        for k, val in cands_v.items():
            print(f"{label:20s} v: {k:28s} equal {float((val == v_t).float().mean()):.6f}")
        sq = torch.sqrt(v_t)
        den_variants = {"sqrt(v)/bc2 + eps": add(div(sq, bc2t), epst),
                        "sqrt(v)*(1/bc2) + eps": add(mul(sq, 1.0 / bc2t), epst),
                        "fma(sqrt v, 1/bc2, eps)": fma(sq, 1.0 / bc2t, epst)}
        for (kd, den), fused in itertools.product(den_variants.items(), (False, True)):
            q = div(m_t, den)
            val = fma(step_size, q, p0) if fused else add(p0, mul(step_size, q))
            name = f"{kd}; " + ("fma(step, m/den, p)" if fused else "p + step*(m/den)")
            print(f"{label:20s} p: {name:52s} equal {float((val == p_t).float().mean()):.6f}")


if __name__ == "__main__":
    main()
