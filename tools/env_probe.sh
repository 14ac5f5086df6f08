#!/bin/bash
# Whole-step time under one GCN_TUNE_* setting after another (tuning build), two runs each.
# Usage: bash tools/env_probe.sh "<config> [--batch N]" <steps> <VAR> <value> [<value> ...]     ("-" = variable unset)
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
cfg=$1; steps=$2; var=$3; shift 3
mkdir -p gpurun_out
for v in "$@"; do
  for rep in 1 2; do
    if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
    timeout -k 10 300 python -u bench.py --config $cfg --steps $steps --warmup 5 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/env_probe.log 2>&1 || { echo "failed $v"; tail -3 gpurun_out/env_probe.log; exit 1; }
    grep -h '"metric"' gpurun_out/env_probe.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $var=$v:', d['ms_per_step'], 'bwd', d['roofline']['avg_launch_us'], 'fwd', d['roofline_fwd']['avg_launch_us'])"
  done
done
