#!/bin/bash
# Grid cap of the wide node kernels (GCN_TUNE_GRID_NW, tuning build) at the H = 300 configuration: whole step, two runs each.
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
mkdir -p gpurun_out
for g in ${GRIDS:-2048 512 256 128}; do
  for rep in 1 2; do
  GCN_TUNE_GRID_NW=$g timeout -k 10 200 python -u bench.py --config synthetic_100n_50c_h300_l30 --steps 3 --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/nwg.log 2>&1 || { echo failed; tail -3 gpurun_out/nwg.log; exit 1; }
  grep -h '"metric"' gpurun_out/nwg.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('NW $g:', d['ms_per_step'])"
  done
done
