bash tools/r02_measure.sh > gpurun_out/r02_measure.log 2>&1 || { tail -20 gpurun_out/r02_measure.log; exit 1; }
cp gpurun_out/r02/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 600 python -u bench.py > gpurun_out/r02/bench_default_run.log 2>&1 || { tail -5 gpurun_out/r02/bench_default_run.log; exit 1; }
tail -1 gpurun_out/r02/bench_default_run.log > gpurun_out/r02/bench_default_run.json
cut -c1-600 gpurun_out/r02/bench_default_run.json
