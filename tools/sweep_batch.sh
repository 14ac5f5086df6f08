#!/bin/bash
# Graph-replay bench of one configuration at several batch sizes under the settings of one tuning variable
# (measurement build of the library).  Usage: bash tools/sweep_batch.sh <config> "<B1 B2 ..>" <VAR> "<v1 v2 ..>" [extra env assignments]
export GCN_UELB_LIB=${GCN_UELB_LIB:-$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so}
cfg=$1; batches=$2; var=$3; vals=$4; shift 4
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out
for B in $batches; do for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  timeout -k 10 200 python -u bench.py --config $cfg --batch $B --steps ${STEPS:-20} --warmup ${WARM:-5} --skip-cpu --skip-beam --skip-sweep --skip-opt > gpurun_out/swb.log 2>&1 || { echo "$cfg B=$B $var=$v failed"; tail -3 gpurun_out/swb.log; exit 1; }
  grep -h '"metric"' gpurun_out/swb.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg B=$B $var=$v $*:', d['ms_per_step'], 'ms/step;', d['value'], 'graphs/s; edge bwd', d['roofline']['avg_launch_us'], 'us')"
done; done
