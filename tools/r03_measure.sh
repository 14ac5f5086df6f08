#!/bin/bash
# End-of-round measurement set (on the GPU box, from the repo root): bash tools/r03_measure.sh
# Per configuration an eager rocprofv3 kernel trace and a graph-replay bench line; the HBM traffic passes of the edge
# kernels for three configurations; instruction-mix passes (counters only, one set per run) at nsfnet_10 and at
# 200 nodes x 100 commodities; the MFMA pass of the H = 300 configuration; the default bench run.  Everything lands under
# gpurun_out/r03/ (copy what is to be kept into profiles/ as r03_*).
out=gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
step() { echo "== $*"; }
for spec in nsfnet_10_commodities:20:50 nsfnet_20_commodities:10:30 default2:10:30 synthetic_200n_100c:2:5 synthetic_100n_50c_h300_l30:2:3; do
  cfg=${spec%%:*}; rest=${spec#*:}; tsteps=${rest%%:*}; bsteps=${rest#*:}
  step "kernel trace $cfg"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_$cfg -- python3 bench.py --config $cfg --steps $tsteps --warmup 2 --skip-cpu --skip-beam --skip-sweep --skip-opt --skip-large --skip-traffic --no-graph > $out/prof_$cfg.log 2>&1 || { echo "trace failed"; tail -5 $out/prof_$cfg.log; exit 1; }
  cp $(ls $out/prof_$cfg/*/*kernel_stats.csv | head -1) $out/kernel_stats_${cfg}_eager.csv
  step "bench $cfg"
  timeout -k 10 300 python -u bench.py --config $cfg --steps $bsteps --warmup 5 --skip-cpu --skip-beam --skip-sweep --skip-large > $out/bench_$cfg.log 2>&1 || { echo "bench failed"; tail -5 $out/bench_$cfg.log; exit 1; }
  grep -h '"metric"' $out/bench_$cfg.log | tail -1 > $out/bench_$cfg.json
  cut -c1-260 $out/bench_$cfg.json
done
step "traffic passes"
bash tools/pmc_traffic.sh nsfnet_10_commodities default2 synthetic_200n_100c > $out/pmc_traffic_run.log 2>&1 || { echo "traffic passes failed"; tail -5 $out/pmc_traffic_run.log; exit 1; }
cp gpurun_out/pmc_traffic.json $out/pmc_traffic.json
tail -10 $out/pmc_traffic_run.log
for cfg in nsfnet_10_commodities synthetic_200n_100c; do
  step "instruction mix $cfg"
  bash tools/pmc_pass.sh r03_sq_$cfg "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "k_edge" $cfg 2 || exit 1
  python tools/pmc_summary.py gpurun_out/pmc_r03_sq_$cfg > $out/pmc_sq_$cfg.txt
done
step "MFMA pass H = 300"
bash tools/pmc_mfma.sh r03_mfma_h300 synthetic_100n_50c_h300_l30 "k_edge_|k_gemm_tn|k_head_" || exit 1
cp gpurun_out/pmc_r03_mfma_h300.txt $out/pmc_mfma_h300.txt
step "default bench run (with the traffic table just taken)"
cp $out/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 600 python -u bench.py > $out/bench_default_run.log 2>&1 || { echo "default run failed"; tail -5 $out/bench_default_run.log; exit 1; }
tail -1 $out/bench_default_run.log > $out/bench_default_run.json
cut -c1-400 $out/bench_default_run.json
echo done
