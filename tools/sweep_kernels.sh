#!/bin/bash
# Eager rocprofv3 kernel trace of one configuration under several settings of one tuning variable; prints the average
# duration of the kernels whose name matches a pattern.  Usage: bash tools/sweep_kernels.sh <config> <batch|0> <pattern> <VAR> "<v1 v2 ..>"
export GCN_UELB_LIB=${GCN_UELB_LIB:-$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so}
cfg=$1; B=$2; pat=$3; var=$4; vals=$5
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in $vals; do
  if [ "$v" = "-" ]; then unset $var; else export $var=$v; fi
  rm -rf gpurun_out/prof_swk
  barg=""; if [ "$B" != "0" ]; then barg="--batch $B"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_swk -- python3 bench.py --config $cfg $barg --steps ${STEPS:-5} --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-opt --no-graph > gpurun_out/prof_swk.log 2>&1 || { echo "failed $var=$v"; tail -3 gpurun_out/prof_swk.log; exit 1; }
  python3 - "$pat" "$var=$v" <<'PY'
import csv, glob, re, sys
f = glob.glob('gpurun_out/prof_swk/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[1], r['Name']):
        print(f"{sys.argv[2]:24s} {r['Name'][:44]:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f}")
PY
done
