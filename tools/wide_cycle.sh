#!/bin/bash
# Wide-path GPU cycle: the parity tests of the wide kernels, then an eager kernel trace of the H = 300 configuration.
# Usage (on the GPU box, from the repo root): bash tools/wide_cycle.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_hip_parity.py -m gpu -q -x -k "wide" --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; grep -E "^(FAILED|ERROR|E  )" gpurun_out/pytest_$tag.log | head -20; exit $rc; fi
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --config synthetic_100n_50c_h300_l30 --steps 2 --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-traffic --no-graph > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof run failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
python tools/prof_summary.py gpurun_out/prof_$tag | head -12
