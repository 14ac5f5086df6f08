#!/bin/bash
# node segment alone, with and without the in-kernel finish (timing probes: results are then wrong)
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in fin nofin; do
  if [ $v = nofin ]; then export GCN_TUNE_NOFIN=1; fi
  GCN_TUNE_MIDSKIP=14 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw6_$v -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --no-graph > gpurun_out/sw6_$v.log 2>&1 || { echo "run $v failed"; tail -3 gpurun_out/sw6_$v.log; exit 1; }
  echo "variant=$v"; python tools/prof_summary.py gpurun_out/sw6_$v | grep -E "k_bwd_mid"
done
