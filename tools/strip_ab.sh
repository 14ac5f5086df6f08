#!/bin/bash
# A/B of the 200-node configuration (eager kernel trace each): graph-convnet-uelb_amd/lib/libgcn_alt_head.so (a build of
# the committed sources), then the tuning library once per argument, each argument a list of GCN_TUNE_* settings
# ("GCN_TUNE_GRID_HB=768", "" for none).  Usage: bash tools/strip_ab.sh <tag> [settings ...]
tag=${1:-x}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
n=0
for g in head "$@"; do
  if [ "$g" = head ]; then lib=$PWD/graph-convnet-uelb_amd/lib/libgcn_alt_head.so; set_=""; else lib=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so; set_="$g"; fi
  echo "== $g"
  ( export GCN_UELB_LIB=$lib; for kv in $set_; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$n -- python3 bench.py --config synthetic_200n_100c --batch 4 --steps 3 --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-opt --skip-traffic --no-graph > gpurun_out/prof_${tag}_$n.log 2>&1 ) || { echo "rocprof run $n failed"; tail -5 gpurun_out/prof_${tag}_$n.log; exit 1; }
  python tools/prof_summary.py gpurun_out/prof_${tag}_$n 4 2>/dev/null | sed -n 1,6p
  n=$((n+1))
done
