#!/bin/bash
# Grid of the head backward kernel (GCN_TUNE_GRID_HB, tuning build) at the small configurations: whole-step time, two runs each.
# (Measured: 512 / 768 / 1024 workgroups within 0.4 % of each other at nsfnet_10 with 20 and 80 graphs and at default2.)
export GCN_UELB_LIB=$PWD/graph-convnet-uelb_amd/lib/libgcn_uelb_hip_tuning.so
for cfg in "nsfnet_10_commodities" "nsfnet_10_commodities --batch 80" "default2"; do
for hb in 512 768 1024; do
  for rep in 1 2; do
  GCN_TUNE_GRID_HB=$hb timeout -k 10 200 python -u bench.py --config $cfg --steps 200 --warmup 20 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic > gpurun_out/hb.log 2>&1 || { echo failed; tail -3 gpurun_out/hb.log; exit 1; }
  grep -h '"metric"' gpurun_out/hb.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg HB $hb:', d['ms_per_step'])"
  done
done
done
