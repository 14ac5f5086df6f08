#!/bin/bash
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
for g in 256 192 128 64; do
  GCN_TUNE_GRID_B=$g GCN_TUNE_GRID_F=$g timeout -k 10 200 python -u bench.py --config synthetic_100n_50c_h300_l30 --steps 2 --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/gridtest.log 2>&1 || { echo failed; tail -3 gpurun_out/gridtest.log; exit 1; }
  grep -h '"metric"' gpurun_out/gridtest.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid $g:', d['ms_per_step'], 'ms/step; edge bwd', d['roofline']['avg_launch_us'], 'us; fwd', d['roofline_fwd']['avg_launch_us'])"
done
