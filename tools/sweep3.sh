#!/bin/bash
run() { echo -n "$* -> "; env "$@" timeout -k 10 120 python bench.py --steps 50 --warmup 10 --skip-cpu --skip-beam 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['loss'])" || exit 1; }
run X=0
run GCN_TUNE_OLDHEAD=1
