#!/bin/bash
# rocprofv3 passes over the kernels of one bench sweep (one --pmc set per run, as the guide prescribes).
# usage: tools/profile_bench.sh <outdir> [bench.py args]
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-extras $*"
REGEX="warp_gram_lists_kernel|lists_tilemask_kernel|gram_lists_finish|warp_recon_grad_kernel|warp_recon_grad_finish|recon_lists_kernel|mu_temporal|adam_epoch"
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --kernel-include-regex "$REGEX" --output-format csv -d "$out/$name" "$@" -- python3 bench.py $ARGS > "$out/$name.log" 2>&1; echo "$name rc=$?"; }
# the bench command itself (default steps / warm-up) under the tracer, then the counter passes on a shorter run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline --no-extras $* > "$out/trace.log" 2>&1; echo "trace rc=$?"
python3 tools/trace_timed_region.py "$out"/trace/*/*_kernel_trace.csv 20 5 "$out/timed_region.json" > "$out/timed_region.txt"
run sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM
run tcp --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run ta --pmc TA_TA_BUSY_sum TA_BUSY_max TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run grbm --pmc GRBM_GUI_ACTIVE GRBM_COUNT
echo done
