#!/bin/bash
# HBM traffic of the dominant edge backward kernel per launch, for the traffic table bench.py reports from:
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (counters only) over a short eager bench of each
# configuration, then tools/pmc_traffic_collect.py writes profiles/pmc_traffic.json with the hash of the library sources.
# Usage (on the GPU box, from the repo root): bash tools/pmc_traffic.sh [config[:batch] ...]
cfgs=${@:-"nsfnet_10_commodities default2 synthetic_200n_100c"}
export PMC_TIMEOUT=${PMC_TIMEOUT:-120}
for c in $cfgs; do
  cfg=${c%%:*}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    echo "pass $cfg $ctr"
    bash tools/pmc_pass.sh traffic_${cfg}_$ctr "$ctr" "k_edge" $cfg 2 || exit 1
  done
done
python tools/pmc_traffic_collect.py $cfgs
