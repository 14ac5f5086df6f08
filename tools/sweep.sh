#!/bin/bash
# bench sweep over tuning environment variables; one line per setting
mkdir -p gpurun_out
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 50 --warmup 10 --skip-cpu --skip-beam 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1; }
run X=0
run GCN_TUNE_GRID_F=512
run GCN_TUNE_GRID_F=1024
run GCN_TUNE_GRID_F=1400
run GCN_TUNE_GRID_F=2048
run GCN_TUNE_GRID_B=512
run GCN_TUNE_GRID_B=1024
run GCN_TUNE_GRID_B=1400
run GCN_TUNE_IT=1
run GCN_TUNE_IT=1 GCN_TUNE_GRID_B=1400
run GCN_TUNE_IT=4
run GCN_TUNE_IT=7
