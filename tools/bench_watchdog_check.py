#!/usr/bin/env python3
"""The safety net of bench.py's N > 1 extra (extras.sharded_footprint_update): rank 1 is made to fail inside it, rank 0
then hangs in the collective; the watchdog (shortened to 25 s here) must still print rank 0's line, with an error entry
in place of the extra, and every rank must leave with exit code 0.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 \
        tools/bench_watchdog_check.py --gpus 2 --backend gloo --frames 500 --steps 2 --warmup 1
"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
_T = threading.Timer
threading.Timer = lambda interval, fn: _T(min(interval, 25.0), fn)
_sleep = time.sleep
time.sleep = lambda s: _sleep(min(s, 40.0))
orig = bench.sharded_footprint_update
def broken(*a, **k):
    if int(os.environ.get("RANK", "0")) == 1:
        raise RuntimeError("injected failure")
    return orig(*a, **k)
bench.sharded_footprint_update = broken
bench.main()
