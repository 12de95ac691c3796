#!/bin/bash
# MFMA counters of K8's axis transform: usage tools/pmc_k8.sh <outdir>
out=$1; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-include-regex "mc_axis_mfma" --output-format csv -d "$out/$name" "$@" -- python3 tools/time_register.py 512 2 128 > "$out/$name.log" 2>&1; echo "$name rc=$?"; }
run mfma --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES
run grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt"
cat "$out/summary.txt"
