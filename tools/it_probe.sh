#!/bin/bash
# Row-chunk length of the edge backward pass (GCN_TUNE_IT, tuning build) against the library's own choice.
# Usage: bash tools/it_probe.sh "<config> [--batch N]" <IT> [<IT> ...]    (0 = the library's choice)
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
cfg=$1; shift
mkdir -p gpurun_out
for it in "$@"; do
  if [ "$it" = 0 ]; then unset GCN_TUNE_IT; else export GCN_TUNE_IT=$it; fi
  timeout -k 10 300 python -u bench.py --config $cfg --steps 3 --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/it_probe.log 2>&1 || { echo "failed $it"; tail -3 gpurun_out/it_probe.log; exit 1; }
  grep -h '"metric"' gpurun_out/it_probe.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('IT $it:', d['ms_per_step'], 'ms/step; edge bwd', d['roofline']['avg_launch_us'], 'us')"
done
