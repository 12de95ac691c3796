#!/usr/bin/env python3
"""profiles/roofline_traffic.json from the counter passes of tools/profile_bench.sh (and tools/profile_k3.sh).

    python tools/make_traffic_json.py <stamp.json> <name>=<summary.json> [...]  [--keep-old]

<summary.json> = the output of tools/pmc_summary.py for one bench geometry; <name> = its key prefix, e.g.
512x512x4000_K100 or 512x512x2x4000_K100.  <stamp.json> = {"file": "hash", ...} written ON THE GPU BOX by the profiled
run itself (`python -c "from dnmf_amd import ops, ...; json.dump(ops.build_stamp(), ...)"`): the hashes of the sources the
profiled library was compiled from.  bench.py hands an entry out only while the library it loads carries the same hashes
for the entry's files.

HBM bytes per launch = FETCH_SIZE (KiB) x 1024 x 2 + WRITE_SIZE (KiB) x 1024: on gfx950 FETCH_SIZE reports half of the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALU_CYCLES = 2.6   # mean issue cost of these kernels' instruction mix (profiles/r02_valu_probe.txt: 2.3 / 4.2 cycles)


def kernels(summary, prefix):
    """{counter: value} summed over the kernels whose name starts with one of `prefix`."""
    out = {}
    for k, v in summary.items():
        name, _, counter = k.rpartition(":")
        if any(name.startswith(p) for p in prefix):
            out[counter] = out.get(counter, 0.0) + v
    return out


def hbm_bytes(c):
    return 2048.0 * c.get("FETCH_SIZE", 0.0) + 1024.0 * c.get("WRITE_SIZE", 0.0)


def issue(c, units):
    """Issue statistics of one kernel; `units` = work items (tiles or voxel-waves) per launch for the per-unit counts."""
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0          # cycles of the launch x SIMDs of the chip
    return {"vector_instructions_per_launch": c["SQ_INSTS_VALU"], "scalar_instructions_per_launch": c["SQ_INSTS_SALU"],
            "vector_instructions_per_unit": c["SQ_INSTS_VALU"] / units, "scalar_instructions_per_unit": c["SQ_INSTS_SALU"] / units,
            "vector_alu_busy_frac_at_2.6_cycles_per_instruction": VALU_CYCLES * c["SQ_INSTS_VALU"] / simd_cycles,
            "waves_per_simd": 4.0 * c["SQ_WAVE_CYCLES"] / simd_cycles,
            "wave_cycles_waiting_frac": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "l2_hit_frac": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), "hbm_bytes": hbm_bytes(c)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stamp = json.load(open(args[0]))
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    entries = {}
    if "--keep-old" in sys.argv and os.path.exists(path):
        entries = json.load(open(path)).get("entries", {})
    for spec in args[1:]:
        name, _, file = spec.partition("=")
        s = json.load(open(file))
        geom = name.split("_K")[0]
        dims = [int(v) for v in geom.split("x")]
        T = dims[-1]
        P = 1
        for d in dims[:-1]:
            P *= d
        zm = 1 if len(dims) == 3 else (2 if dims[2] == 2 else 3)
        lists = ["warp_gram_lists.hip", "common.hpp"] + (["warp_gram_lists_z.hip"] if zm > 1 else [])
        p1 = kernels(s, [f"warp_gram_lists_kernel<{zm}, "])
        one = {k.rpartition(":")[0] for k in s if k.startswith(f"warp_gram_lists_kernel<{zm}, ")}
        tm = kernels(s, ["lists_tilemask_kernel"])
        if p1:
            entries[name + "_lists"] = {"value": hbm_bytes(p1) + hbm_bytes(tm), "files": lists,
                                        "what": "HBM bytes per K3n call: its launches " + ", ".join(sorted(one)) + " + lists_tilemask_kernel"}
            tiles = T * (P / 256.0)
            valu = {}
            for kname in sorted(one):
                c = kernels(s, [kname])
                passno = kname.rstrip(">").split(",")[-1].strip()
                valu[{"0": "one_kernel_form", "1": "short_list_pass", "2": "long_list_pass"}.get(passno, kname)] = issue(c, tiles)
            valu["note"] = ("per unit = per 256-voxel tile and wave; counter passes serialise the two launches, which otherwise "
                            "run side by side: busy fractions and waves per SIMD are those of each kernel alone")
            entries[name + "_lists_valu"] = {"value": valu, "files": lists}
        k2 = kernels(s, ["warp_recon_grad_kernel<"])
        if k2:
            entries[geom + "_K2"] = {"value": hbm_bytes(k2), "files": ["warp_recon_grad.hip", "common.hpp"],
                                     "what": "HBM bytes per K2 launch", "issue": issue(k2, T * P / 64.0)}
        rl = kernels(s, ["recon_lists_kernel<"])
        if rl:
            entries[geom + "_recon_lists"] = {"value": hbm_bytes(rl), "files": ["recon_lists.hip", "common.hpp"],
                                             "what": "HBM bytes per list-reconstruction launch"}
        for kern, key, files in (("warp_gram_kernel<", name, ["warp_gram_rhs.hip", "common.hpp"]),
                                 ("warp_gram_lt_kernel<", name + "_sparse", ["warp_gram_sparse.hip", "common.hpp"])):
            c = kernels(s, [kern])
            if c and "FETCH_SIZE" in c:
                entries[key] = {"value": hbm_bytes(c), "files": files, "what": "HBM bytes per launch of " + kern.rstrip("<")}
    for e in entries.values():          # the hashes of an entry's files at measurement time travel with the entry
        if "sources" not in e:
            e["sources"] = {f: stamp.get(f) for f in e["files"]}
    out = {"_note": __doc__.strip().split("\n\n")[-1], "entries": entries}
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print("wrote", path, "with", len(entries), "entries")


if __name__ == "__main__":
    main()
