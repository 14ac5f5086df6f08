#!/bin/bash
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 50 --warmup 10 --skip-cpu --skip-beam 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'])" || exit 1; }
for p in 0 1 2 4 8 6 7 15; do run GCN_TUNE_PROBE=$p; done
