#!/bin/bash
# Same-box A/B of two builds of the library (graph-replay bench), alternating.
# Usage: bash tools/ab_libs.sh <config> <steps> <libA.so> <libB.so>   (paths relative to the repo root; "-" = the product library)
cfg=$1; steps=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset GCN_UELB_LIB; else export GCN_UELB_LIB=$PWD/$lib; fi
    timeout -k 10 200 python -u bench.py --config $cfg --steps $steps --warmup 5 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/ab_lib.log 2>&1 || { echo "failed $lib"; tail -3 gpurun_out/ab_lib.log; exit 1; }
    grep -h '"metric"' gpurun_out/ab_lib.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $lib:', d['ms_per_step'], 'ms/step; edge bwd', d['roofline']['avg_launch_us'], 'us')"
  done
done
