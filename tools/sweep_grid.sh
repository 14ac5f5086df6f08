#!/bin/bash
# persistent-grid sweep of the two edge kernels (eager rocprof runs); 512 workgroups = 2 per CU is the optimum at nsfnet_10
export TMPDIR=/tmp
mkdir -p gpurun_out
for g in 256 384 512 768 1024; do
  GCN_TUNE_GRID_B=$g GCN_TUNE_GRID_F=$g timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/swg_$g -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --skip-sweep --no-graph > gpurun_out/swg_$g.log 2>&1 || { echo "run $g failed"; tail -3 gpurun_out/swg_$g.log; exit 1; }
  echo "grid=$g"; python tools/prof_summary.py gpurun_out/swg_$g | grep -E "per step|k_edge_bwd3|k_edge_fwd3<4, 1"
done
