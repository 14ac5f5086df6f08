#!/bin/bash
# MFMA utilisation counters (one pass, counters only) for a bench configuration.
# Usage: bash tools/pmc_mfma.sh <tag> <config> [kernel regex]
tag=$1; cfg=$2; rx=${3:-k_}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAVES --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --config $cfg --steps 1 --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-traffic --no-graph > gpurun_out/pmc_$tag.log 2>&1 || { echo "pmc pass failed"; tail -5 gpurun_out/pmc_$tag.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_$tag.txt
echo ok
