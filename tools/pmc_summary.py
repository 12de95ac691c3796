#!/usr/bin/env python3
"""Average the per-dispatch counter values of the rocprofv3 passes written by tools/profile_k3.sh.

    python tools/pmc_summary.py <dir> [out.json]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    out = {}
    for f in sorted(glob.glob(os.path.join(d, "*", "*", "*_counter_collection.csv"))):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kern = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("dnmf::", "")
            agg[kern + ":" + r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k] = sum(v) / len(v)
    for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
        for r in csv.DictReader(open(f)):
            out["kernel_avg_ms:" + r["Name"][:60]] = float(r["AverageNs"]) / 1e6
    for k in sorted(out):
        print(f"{k:45s} {out[k]:.6g}")
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
