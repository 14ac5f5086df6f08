#!/bin/bash
# Eager kernel trace of the headline configuration at 320 graphs per batch (the two-block kernel regime at NSFNet size).
# Usage (on the GPU box, from the repo root): bash tools/trace_b320.sh
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b320 -- python3 bench.py --config nsfnet_10_commodities --batch 320 --steps 5 --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-opt --skip-traffic --no-graph > gpurun_out/prof_b320.log 2>&1 || { echo failed; tail -5 gpurun_out/prof_b320.log; exit 1; }
python tools/prof_summary.py gpurun_out/prof_b320 8 | head -16
