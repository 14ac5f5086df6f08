#!/bin/bash
# Graph-replay bench of the configurations other than the headline one.  Usage: bash tools/other_configs.sh <tag>
tag=${1:-x}
mkdir -p gpurun_out
for cfg in default2 nsfnet_20_commodities synthetic_200n_100c; do
  steps=30; [ $cfg = synthetic_200n_100c ] && steps=5
  timeout -k 10 300 python -u bench.py --config $cfg --steps $steps --warmup 3 --skip-cpu --skip-beam --skip-sweep > gpurun_out/bench_${tag}_$cfg.log 2>&1 || { echo "$cfg failed"; tail -5 gpurun_out/bench_${tag}_$cfg.log; exit 1; }
  grep -h '"metric"' gpurun_out/bench_${tag}_$cfg.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], 'graphs/s', d['ms_per_step'], 'ms/step', d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['unit'])"
done
