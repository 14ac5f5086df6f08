#!/bin/bash
# One rocprofv3 counter pass over the eager bench (own run: counters only, no tracing).
# Usage: bash tools/pmc_pass.sh <tag> "<COUNTER ...>" [kernel regex] [config] [steps]
tag=$1; ctrs=$2; rx=${3:-k_edge}; cfg=${4:-nsfnet_10_commodities}; steps=${5:-3}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -s KILL ${PMC_TIMEOUT:-90} rocprofv3 --pmc $ctrs --kernel-include-regex "$rx" --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --config $cfg --steps $steps --warmup 1 --skip-cpu --skip-beam --skip-sweep --skip-large --skip-traffic --no-graph > gpurun_out/pmc_$tag.log 2>&1 || { echo "pmc pass failed"; tail -5 gpurun_out/pmc_$tag.log; exit 1; }
echo ok
