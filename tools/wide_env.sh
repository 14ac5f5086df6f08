#!/bin/bash
# Kernel trace of the H = 300 configuration under several settings of one tuning variable.
# Usage: bash tools/wide_env.sh <tag> <VAR> <value> [<value> ...]
tag=$1; var=$2; shift 2
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "$@"; do
  export $var=$v
  echo "== $var=$v"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$v -- python3 bench.py --config synthetic_100n_50c_h300_l30 --steps 2 --warmup 1 --skip-cpu --skip-beam --skip-sweep --no-graph > gpurun_out/prof_${tag}_$v.log 2>&1 || { echo "rocprof run $v failed"; tail -5 gpurun_out/prof_${tag}_$v.log; exit 1; }
  python tools/prof_summary.py gpurun_out/prof_${tag}_$v 2>/dev/null | sed -n 1,4p
done
