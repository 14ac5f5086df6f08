#!/bin/bash
# One GPU cycle: parity tests, then (only if they pass) an eager rocprof kernel trace and a graph-replay bench.
# Usage (on the GPU box, from the repo root): bash tools/gpu_cycle.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -4 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; grep -E "^(FAILED|ERROR|E  )" gpurun_out/pytest_$tag.log | head -20; exit $rc; fi
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 20 --warmup 5 --skip-cpu --skip-beam --skip-sweep --no-graph > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof run failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
timeout -k 10 300 python -u bench.py --steps 50 --warmup 10 --skip-cpu > gpurun_out/bench_$tag.log 2>&1 || { echo "bench failed"; tail -5 gpurun_out/bench_$tag.log; exit 1; }
tail -1 gpurun_out/bench_$tag.log | cut -c1-2000
