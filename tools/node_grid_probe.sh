#!/bin/bash
# Grid cap of the strip node kernels (GCN_TUNE_GRID_N, tuning build): whole-step time at several sizes, two runs each.
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
mkdir -p gpurun_out
for cfg in "nsfnet_10_commodities --batch 320" "nsfnet_10_commodities --batch 80" ${ALLCFG:+"nsfnet_10_commodities" "synthetic_200n_100c"}; do
for g in ${GRIDS:-2048 1024 768 512}; do
  for rep in 1 2; do
  GCN_TUNE_GRID_N=$g timeout -k 10 200 python -u bench.py --config $cfg --steps 30 --warmup 5 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/ng.log 2>&1 || { echo failed; tail -3 gpurun_out/ng.log; exit 1; }
  grep -h '"metric"' gpurun_out/ng.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg N $g:', d['ms_per_step'])"
  done
done
done
