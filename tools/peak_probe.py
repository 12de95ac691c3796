#!/usr/bin/env python3
"""What this box actually delivers, next to the spec figures the rooflines are quoted against ('confirm
with a micro-benchmark on the box and report against both'): HBM copy / fill bandwidth and the fp32 GEMM rate of the
vendor library."""
import json
import sys
import time

import torch


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    out = {"device": torch.cuda.get_device_name(0)}
    n = 1 << 30  # 4 GiB of fp32
    x = torch.empty(n, device="cuda").normal_()
    y = torch.empty_like(x)
    t = timed(lambda: y.copy_(x))
    out["hbm_copy_GBps_read_plus_write"] = 8.0 * n / t / 1e9
    t = timed(lambda: y.fill_(1.0))
    out["hbm_fill_GBps_write"] = 4.0 * n / t / 1e9
    t = timed(lambda: x.sum())
    out["hbm_reduce_GBps_read"] = 4.0 * n / t / 1e9
    del x, y
    m = 8192
    a = torch.randn(m, m, device="cuda")
    b = torch.randn(m, m, device="cuda")
    torch.backends.cuda.matmul.allow_tf32 = False
    t = timed(lambda: a @ b, n=5)
    out["fp32_gemm_TFLOPs_8192"] = 2.0 * m ** 3 / t / 1e12
    out["spec"] = {"hbm_GBps": 8000, "fp32_matrix_TFLOPs": 157.3}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
