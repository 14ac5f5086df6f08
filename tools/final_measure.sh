#!/bin/bash
# End-of-round measurement: parity tests, eager kernel trace, graph-replay bench, default bench (with the CPU baseline),
# and the counter passes for the two edge kernels (each counter set in a run of its own).
# Usage (on the GPU box, from the repo root): bash tools/final_measure.sh <tag>
tag=${1:-final}
set -o pipefail
bash tools/gpu_cycle.sh $tag || exit 1
timeout -k 10 400 python -u bench.py > gpurun_out/bench_default_$tag.log 2>&1 || { echo "default bench failed"; tail -5 gpurun_out/bench_default_$tag.log; exit 1; }
tail -1 gpurun_out/bench_default_$tag.log | cut -c1-300
bash tools/pmc_pass.sh ${tag}_fetch "FETCH_SIZE" k_edge || exit 1
bash tools/pmc_pass.sh ${tag}_write "WRITE_SIZE" k_edge || exit 1
bash tools/pmc_pass.sh ${tag}_sq "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" k_edge || exit 1
for p in fetch write sq; do python tools/pmc_summary.py gpurun_out/pmc_${tag}_$p > gpurun_out/pmc_${tag}_$p.txt; done
cat gpurun_out/pmc_${tag}_fetch.txt gpurun_out/pmc_${tag}_write.txt | grep -A2 "k_edge_bwd3"
