#!/bin/bash
# Everything profiles/ needs for one round, on the GPU box: the source stamp of the library, the rocprofv3 kernel trace and
# counter passes of the bench at Z = 1 and Z = 2 (tools/profile_bench.sh), fetch / write passes of the dense and the
# zero-skipping Gram kernels, kernel stats of the --with-spatial sweep.
# usage: tools/profile_round.sh <outdir>
out=$1
mkdir -p "$out"
python3 -c "import json; from dnmf_amd import ops; json.dump(ops.build_stamp(), open('$out/stamp.json', 'w'), indent=1, sort_keys=True)" || exit 1
bash tools/profile_bench.sh "$out/z1" || exit 1
python3 tools/pmc_summary.py "$out/z1" "$out/z1/summary.json" > "$out/z1/summary.txt"
bash tools/profile_bench.sh "$out/z2" --depth 2 || exit 1
python3 tools/pmc_summary.py "$out/z2" "$out/z2/summary.json" > "$out/z2/summary.txt"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in dense sparse; do
  extra=""; [ $v = sparse ] && extra="--sparse"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --kernel-include-regex "warp_gram.*kernel" --output-format csv -d "$out/k3_$v/$c" --pmc $c -- python3 tools/run_k3.py --frames 4000 --reps 2 $extra > "$out/k3_${v}_$c.log" 2>&1 || exit 1
  done
  python3 tools/pmc_summary.py "$out/k3_$v" "$out/k3_$v/summary.json" > "$out/k3_$v/summary.txt"
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/spatial" -- python3 bench.py --with-spatial --no-cpu-baseline --steps 5 --warmup 2 > "$out/spatial.log" 2>&1 || exit 1
echo profile_round done
