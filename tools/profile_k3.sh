#!/bin/bash
# rocprofv3 counter passes over the K3 Gram kernel (one --pmc set per run, as the guide prescribes).
# usage: tools/profile_k3.sh <outdir> [run_k3.py args]
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="$*"
run() { name=$1; shift; timeout -k 10 150 rocprofv3 --kernel-include-regex "warp_gram.*kernel|gram_.*finish" --output-format csv -d "$out/$name" "$@" -- python3 tools/run_k3.py $ARGS > "$out/$name.log" 2>&1; }
run trace --kernel-trace --stats
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA
run sq2 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM
run ta --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tcp2 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TD_TD_BUSY_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
echo done
