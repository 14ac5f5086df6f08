#!/bin/bash
export TMPDIR=/tmp
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
for spec in ${SPECS:-1024:512 2048:1024 1536:768}; do
  gh=${spec%%:*}; ghb=${spec#*:}
  GCN_TUNE_GRID_H=$gh GCN_TUNE_GRID_HB=$ghb timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_hs_$gh_$ghb -- python3 bench.py --config synthetic_200n_100c --batch 4 --steps 2 --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-opt --skip-traffic --no-graph > gpurun_out/prof_hs.log 2>&1 || { echo failed; tail -3 gpurun_out/prof_hs.log; exit 1; }
  echo "grid fwd $gh bwd $ghb"; python tools/prof_summary.py gpurun_out/prof_hs_$gh_$ghb 2>/dev/null | grep "k_head"
  rm -rf gpurun_out/prof_hs_$gh_$ghb
done
