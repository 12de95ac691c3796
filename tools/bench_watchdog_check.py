#!/usr/bin/env python3
"""The safety net of bench.py's N > 1 extra (--exchange-extra: extras.sharded_footprint_update): rank 1 is made to fail
inside it while rank 0 hangs in the collective.  Expected: rank 1 posts its error, the watchdogs of both ranks see it
within a second, rank 0 prints the line with the REAL error text in place of the extra, and every rank leaves with exit
code 3 (torch.distributed.run then reports the failure: a broken collective must not look like a clean run).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tools/bench_watchdog_check.py --gpus 2 --backend gloo --frames 500 --steps 2 --warmup 1 --exchange-extra
    echo $?    # non-zero; the JSON line on stdout carries "sharded_footprint_update": {"error": "rank 1: RuntimeError('injected failure')"}
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

orig = bench.sharded_footprint_update


def broken(*a, **k):
    if int(os.environ.get("RANK", "0")) == 1:
        raise RuntimeError("injected failure")
    return orig(*a, **k)


bench.sharded_footprint_update = broken
bench.main()
