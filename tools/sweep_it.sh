#!/bin/bash
# Graph-replay bench of one configuration under several row-chunk lengths of the edge backward pass.
# Usage: bash tools/sweep_it.sh <config> <it> [<it> ...]
cfg=$1; shift
mkdir -p gpurun_out
for g in "$@"; do
  GCN_TUNE_IT=$g timeout -k 10 200 python -u bench.py --config $cfg --steps 50 --warmup 10 --skip-cpu --skip-beam --skip-opt > gpurun_out/swi_${cfg}_$g.log 2>&1 || { echo "$cfg $g failed"; tail -3 gpurun_out/swi_${cfg}_$g.log; exit 1; }
  grep -h '"metric"' gpurun_out/swi_${cfg}_$g.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg IT $g:', d['ms_per_step'], 'ms/step', d['value'], 'graphs/s', d['roofline']['avg_launch_us'])"
done
