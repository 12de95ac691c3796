#!/bin/bash
# rocprofv3 counter passes over the K3 Gram kernel (one --pmc set per run, as the guide prescribes).
# usage: tools/profile_k3.sh <outdir> [run_k3.py args]
set -e
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; rocprofv3 --kernel-include-regex warp_gram --output-format csv -d "$out/$name" "$@" -- python3 tools/run_k3.py $ARGS > "$out/$name.log" 2>&1; }
ARGS="$*"
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
echo done
