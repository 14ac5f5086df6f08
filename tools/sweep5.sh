#!/bin/bash
# timing probe of the merged backward launch with and without the in-kernel finish
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in fin nofin; do
  if [ $v = nofin ]; then export GCN_TUNE_NOFIN=1; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw5_$v -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --no-graph > gpurun_out/sw5_$v.log 2>&1 || { echo "run $v failed"; tail -3 gpurun_out/sw5_$v.log; exit 1; }
  echo "variant=$v"; python tools/prof_summary.py gpurun_out/sw5_$v | grep -E "per step|k_bwd_mid|k_wgrad|k_reduce"
done
