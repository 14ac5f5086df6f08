#!/bin/bash
# A/B of library builds on the H = 300 configuration (eager kernel trace each): the in-tree library, then every
# graph-convnet-uelb_amd/lib/libgcn_alt*.so (probe builds; their results are not checked).  Usage: bash tools/wide_ab.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 400 python -u -m pytest tests/test_hip_parity.py -m gpu -q -x -k "wide" --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_$tag.log
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; grep -E "^(FAILED|ERROR|E  )" gpurun_out/pytest_$tag.log | head -20; exit $rc; fi
n=0
for lib in "" graph-convnet-uelb_amd/lib/libgcn_alt*.so; do
  if [ -n "$lib" ]; then [ -f "$lib" ] || continue; export GCN_UELB_LIB=$PWD/$lib; fi
  echo "== ${lib:-main}"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$n -- python3 bench.py --config synthetic_100n_50c_h300_l30 --steps 2 --warmup 1 --skip-cpu --skip-beam --skip-sweep --no-graph > gpurun_out/prof_${tag}_$n.log 2>&1 || { echo "rocprof run $n failed"; tail -5 gpurun_out/prof_${tag}_$n.log; exit 1; }
  python tools/prof_summary.py gpurun_out/prof_${tag}_$n 2>/dev/null | sed -n 1,4p
  n=$((n+1))
done
