#!/bin/bash
# Copies the files of the last tools/r03_measure.sh run (gpurun_out/r03/) into profiles/ under their r03_* names.
set -e
cd "$(dirname "$0")/.."
for c in default2 nsfnet_10_commodities nsfnet_20_commodities synthetic_100n_50c_h300_l30 synthetic_200n_100c; do
  cp gpurun_out/r03/kernel_stats_${c}_eager.csv profiles/r03_kernel_stats_${c}_eager.csv
  cp gpurun_out/r03/bench_$c.json profiles/r03_bench_$c.json
done
cp gpurun_out/r03/pmc_sq_nsfnet_10_commodities.txt profiles/r03_pmc_sq_nsfnet_10_commodities.txt
cp gpurun_out/r03/pmc_sq_synthetic_200n_100c.txt profiles/r03_pmc_sq_synthetic_200n_100c.txt
cp gpurun_out/r03/pmc_mfma_h300.txt profiles/r03_pmc_mfma_h300.txt
cp gpurun_out/r03/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/r03/pmc_traffic.json profiles/r03_pmc_traffic.json
cp gpurun_out/r03/bench_default_run.json profiles/r03_bench_default_run.json
cp gpurun_out/r03/bench_default_run.log profiles/r03_bench_default_run.log
