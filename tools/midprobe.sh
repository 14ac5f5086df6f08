#!/bin/bash
# Time of k_bwd_mid with segments of the launch dropped (tuning build; results of the step are wrong under a probe).
# Usage: bash tools/midprobe.sh <config> <mask> [<mask> ...]   mask bits: 1 node backward, 2 edge product, 4 node products, 8 reduce jobs
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
export TMPDIR=/tmp
cfg=$1; shift
for m in "$@"; do
  export GCN_TUNE_MIDPROBE=$m
  rm -rf gpurun_out/prof_mid_$m
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mid_$m -- python3 bench.py --config $cfg --steps 10 --warmup 3 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic --no-graph > gpurun_out/prof_mid_$m.log 2>&1 || { echo "failed $m"; tail -3 gpurun_out/prof_mid_$m.log; exit 1; }
  f=$(find gpurun_out/prof_mid_$m -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $m <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if r['Name'].startswith('void k_bwd_mid') or r['Name'].startswith('void k_edge_bwd3'):
        print('mask', sys.argv[2], r['Name'][:24], 'avg %.2f us' % (float(r['AverageNs'])/1e3))
PY
done
