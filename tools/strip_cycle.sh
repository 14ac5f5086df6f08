#!/bin/bash
# Strip-path GPU cycle: the parity tests of the strip kernels, then an eager kernel trace of the 200-node configuration.
# Usage (on the GPU box, from the repo root): bash tools/strip_cycle.sh <tag> [batch]
tag=${1:-x}; batch=${2:-4}
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests/test_hip_parity.py -m gpu -q -x -k "not wide" --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; grep -E "^(FAILED|ERROR|E  )" gpurun_out/pytest_$tag.log | head -20; exit $rc; fi
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --config synthetic_200n_100c --batch $batch --steps 3 --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-opt --no-graph > gpurun_out/prof_$tag.log 2>&1 || { echo "rocprof run failed"; tail -5 gpurun_out/prof_$tag.log; exit 1; }
tail -1 gpurun_out/prof_$tag.log | cut -c1-300
python tools/prof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_$tag.txt 2>/dev/null
head -14 gpurun_out/prof_$tag.txt
