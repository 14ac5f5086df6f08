#!/bin/bash
# Graph-replay bench of one configuration under several settings of one tuning variable.
# The GCN_TUNE_* variables are read only by the measurement build of the library (tools/build_tuning_lib.sh), which this
# script selects unless GCN_UELB_LIB already names another build.
# Usage: bash tools/sweep_var.sh <config> <VAR> <value> [<value> ...]
export GCN_UELB_LIB=${GCN_UELB_LIB:-$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so}
cfg=$1; var=$2; shift 2
mkdir -p gpurun_out
for g in "$@"; do
  export $var=$g
  timeout -k 10 200 python -u bench.py --config $cfg --steps ${STEPS:-50} --warmup ${WARM:-10} --skip-cpu --skip-beam --skip-sweep --skip-opt > gpurun_out/swv_${cfg}_${var}_$g.log 2>&1 || { echo "$cfg $var=$g failed"; tail -3 gpurun_out/swv_${cfg}_${var}_$g.log; exit 1; }
  grep -h '"metric"' gpurun_out/swv_${cfg}_${var}_$g.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $var=$g:', d['ms_per_step'], 'ms/step', d['value'], 'graphs/s')"
done
