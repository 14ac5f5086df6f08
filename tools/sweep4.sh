#!/bin/bash
# slice-count sweep of the weight-gradient products: eager rocprof runs, k_wgrad average per setting
export TMPDIR=/tmp
mkdir -p gpurun_out
for s in 160 256 384 512; do
  GCN_TUNE_SLICES=$s timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw4_$s -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --no-graph > gpurun_out/sw4_$s.log 2>&1 || { echo "run $s failed"; tail -3 gpurun_out/sw4_$s.log; exit 1; }
  echo "slices=$s"; python tools/prof_summary.py gpurun_out/sw4_$s | grep -E "per step|k_bwd_mid|k_wgrad|k_reduce"
done
