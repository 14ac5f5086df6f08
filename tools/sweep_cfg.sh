#!/bin/bash
# Graph-replay bench of one configuration under several edge-kernel grid sizes.  Usage: bash tools/sweep_cfg.sh <config> <grid> [<grid> ...]
cfg=$1; shift
mkdir -p gpurun_out
for g in "$@"; do
  GCN_TUNE_GRID_B=$g GCN_TUNE_GRID_F=$g timeout -k 10 200 python -u bench.py --config $cfg --steps 30 --warmup 5 --skip-cpu --skip-beam --skip-opt > gpurun_out/swc_${cfg}_$g.log 2>&1 || { echo "$cfg $g failed"; tail -3 gpurun_out/swc_${cfg}_$g.log; exit 1; }
  grep -h '"metric"' gpurun_out/swc_${cfg}_$g.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg grid $g:', d['ms_per_step'], 'ms/step', d['value'], 'graphs/s')"
done
