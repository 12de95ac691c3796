#!/usr/bin/env python3
"""Per-kernel average duration over the TIMED steps of a bench run, from a rocprofv3 --kernel-trace CSV.

    python tools/trace_timed_region.py <..._kernel_trace.csv> <steps> <warmup> [out.json]

`rocprofv3 --stats` averages over every dispatch of the process, including the warm-up steps, whose launches run
30-70 % slower (first touch of freshly allocated buffers, clock ramp).  bench.py times the last <steps> sweeps only;
this script averages the last <steps> x (dispatches per sweep) dispatches of every kernel that runs in each sweep, so
the two can be compared.
"""
import collections
import csv
import json
import sys


def main():
    path, steps, warmup = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        by[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    out = {}
    for name, rows in by.items():
        if "dnmf::" not in name or len(rows) % (steps + warmup):
            continue  # not launched the same number of times in every sweep (set-up kernels, the extra dense launch)
        rows.sort()
        per_step = len(rows) // (steps + warmup)
        last = [d for _, d in rows[-per_step * steps:]]
        short = name.split("(")[0].replace("void ", "")
        out[short] = {"launches_in_timed_region": len(last), "avg_ms": sum(last) / len(last) / 1e6,
                      "min_ms": min(last) / 1e6, "max_ms": max(last) / 1e6, "all_launches": len(rows),
                      "avg_ms_all_launches": sum(d for _, d in rows) / len(rows) / 1e6}
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches_in_timed_region"]):
        print(f"{k:60s} {v['launches_in_timed_region']:4d} x {v['avg_ms']:.4f} ms  (all {v['all_launches']} launches: {v['avg_ms_all_launches']:.4f})")
    if len(sys.argv) > 4:
        json.dump(out, open(sys.argv[4], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
