#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly file (hipcc -save-temps).

    python tools/isa_loops.py FILE.s KERNEL_SUBSTRING [--all]

For every backward branch of the kernel whose mangled name contains KERNEL_SUBSTRING: the number of vector-ALU,
vector-memory, LDS and scalar instructions between the branch target and the branch, and the most frequent opcodes.
Half-rate vector opcodes (conversions, 24-bit multiplies, v_med3, DPP, compares / selects: tools/valu_probe.hip)
are counted separately, so `units` = full-rate + 1.85 x half-rate approximates issue cycles / 2.25.
"""
import re
import sys
from collections import Counter

HALF = ("v_cvt_", "v_floor_", "v_med3_", "v_min3_", "v_max3_", "v_mad_u32_u24", "v_mad_i32_i24", "v_mul_u32_u24",
        "v_mul_i32_i24", "v_mul_lo_", "v_mul_hi_", "v_cmp_", "v_cndmask_", "v_lshl_add_", "v_add_lshl_", "v_lshl_or_",
        "v_and_or_", "v_or3_", "v_add3_", "v_bfe_", "v_readlane", "v_readfirstlane", "v_writelane", "v_mbcnt", "v_pk_",
        "v_fract_", "v_trunc_", "v_rndne_", "v_ceil_")
QUARTER = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_", "v_div_")


def main():
    path, key = sys.argv[1], sys.argv[2]
    txt = open(path).read()
    found = False
    for m in re.finditer(r"^(\S*%s\S*):" % re.escape(key), txt, re.M):
        name = m.group(1)
        if name.startswith(".") or "$" in name:
            continue
        end = txt.find(".Lfunc_end", m.end())
        body = txt[m.end():end].split("\n")
        found = True
        print("==", name)
        ins = []
        labels = {}
        for line in body:
            s = line.strip()
            lm = re.match(r"^(\.LBB\d+_\d+):", s)
            if lm:
                labels[lm.group(1)] = len(ins)
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            ins.append(s.split(";")[0].strip())
        tot = Counter(i.split()[0] for i in ins)
        print("   whole kernel: %d instructions, VALU %d" % (len(ins), sum(v for k, v in tot.items() if k.startswith("v_"))))
        for n, i in enumerate(ins):
            bm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", i) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", i)
            if bm and bm.group(1) in labels and labels[bm.group(1)] <= n:
                seg = ins[labels[bm.group(1)]:n + 1]
                c = Counter(x.split()[0] for x in seg)
                dpp = sum(1 for x in seg if "row_" in x or "quad_perm" in x or "wave_" in x)
                valu = sum(v for k, v in c.items() if k.startswith("v_"))
                half = sum(v for k, v in c.items() if k.startswith(HALF)) + sum(
                    1 for x in seg if ("row_" in x or "quad_perm" in x) and not x.split()[0].startswith(HALF))
                quarter = sum(v for k, v in c.items() if k.startswith(QUARTER))
                vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
                lds = sum(v for k, v in c.items() if k.startswith("ds_"))
                salu = sum(v for k, v in c.items() if k.startswith("s_"))
                units = (valu - half - quarter) + 1.85 * half + 3.6 * quarter
                print("   loop %s: %d instr | VALU %d (half-rate %d, quarter %d, dpp %d) units %.0f | VMEM %d LDS %d SALU %d"
                      % (bm.group(1), len(seg), valu, half, quarter, dpp, units, vmem, lds, salu))
                if "--all" in sys.argv or True:
                    print("      " + ", ".join("%s %d" % kv for kv in c.most_common(28)))
    if not found:
        print("no kernel matching", key)


if __name__ == "__main__":
    main()
