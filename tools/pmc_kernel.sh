#!/bin/bash
# rocprofv3 counter passes (one --pmc set per run, as the guide prescribes) over the kernels matching a regex of ANY command.
# usage: tools/pmc_kernel.sh <outdir> <kernel-regex> <python script and args ...>     (runs `python3 <script and args>`)
out=$1; regex=$2; shift 2
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-include-regex "$regex" --output-format csv -d "$out/$name" "$@" -- python3 $CMD > "$out/$name.log" 2>&1; echo "$name rc=$?"; }
CMD="$*"
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU || exit 1
run sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM || exit 1
run ta --pmc TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum || exit 1
run fetch --pmc FETCH_SIZE || exit 1
run write --pmc WRITE_SIZE || exit 1
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum || exit 1
run grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
python3 tools/pmc_summary.py "$out" "$out/summary.json" > "$out/summary.txt"
echo done
