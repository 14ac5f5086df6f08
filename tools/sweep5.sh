#!/bin/bash
# timing probes of the merged backward launch: segments switched off one at a time (results are then wrong)
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in 0 1 2 4 8 14; do
  GCN_TUNE_MIDSKIP=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw5_$v -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --no-graph > gpurun_out/sw5_$v.log 2>&1 || { echo "run $v failed"; tail -3 gpurun_out/sw5_$v.log; exit 1; }
  echo "skip=$v"; python tools/prof_summary.py gpurun_out/sw5_$v | grep -E "k_bwd_mid"
done
