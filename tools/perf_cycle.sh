#!/bin/bash
# Performance cycle without the parity tests: eager rocprofv3 kernel trace (per-kernel table) and graph-replay bench.
# Usage (on the GPU box, from the repo root): bash tools/perf_cycle.sh <tag> [config] [trace steps] [extra bench args]
tag=${1:-x}; cfg=${2:-nsfnet_10_commodities}; steps=${3:-20}; if [ $# -ge 3 ]; then shift 3; else shift $#; fi
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --config $cfg --steps $steps --warmup 5 --skip-cpu --skip-beam --skip-sweep --skip-opt --no-graph "$@" > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof run failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
python tools/prof_summary.py gpurun_out/prof_$tag $((steps + 5 + 3 + 20)) | head -16
timeout -k 10 300 python -u bench.py --config $cfg --steps 50 --warmup 10 --skip-cpu --skip-beam --skip-sweep "$@" > gpurun_out/bench_$tag.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'graphs/s', d['value'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'us')"
