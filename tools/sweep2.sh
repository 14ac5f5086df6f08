#!/bin/bash
# Graph-replay bench of one configuration under combinations of two tuning variables; prints the step time and the
# edge backward kernel's average launch time.
# Usage: bash tools/sweep2.sh <config> <VAR1> "<v1 v2 ..>" <VAR2> "<w1 w2 ..>"
export GCN_UELB_LIB=${GCN_UELB_LIB:-$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so}
cfg=$1; var1=$2; vals1=$3; var2=$4; vals2=$5
mkdir -p gpurun_out
for a in $vals1; do for b in $vals2; do
  export $var1=$a; export $var2=$b
  timeout -k 10 200 python -u bench.py --config $cfg --steps ${STEPS:-30} --warmup ${WARM:-5} --skip-cpu --skip-beam --skip-sweep --skip-opt > gpurun_out/sw2.log 2>&1 || { echo "$cfg $var1=$a $var2=$b failed"; tail -3 gpurun_out/sw2.log; exit 1; }
  grep -h '"metric"' gpurun_out/sw2.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $var1=$a $var2=$b:', d['ms_per_step'], 'ms/step; edge bwd', d['roofline']['avg_launch_us'], 'us')"
done; done
