set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -q -x --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/pytest_s3c.log 2>&1 || { tail -30 gpurun_out/pytest_s3c.log; exit 1; }
tail -2 gpurun_out/pytest_s3c.log
for cfg in synthetic_200n_100c; do
timeout -k 10 300 python -u bench.py --config $cfg --steps 10 --warmup 3 --skip-cpu --skip-beam --skip-sweep --skip-opt > gpurun_out/b_$cfg.log 2>&1
tail -1 gpurun_out/b_$cfg.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['ms_per_step'], 'ms/step', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done
timeout -k 10 300 python -u bench.py --steps 50 --warmup 10 --skip-cpu --skip-beam > gpurun_out/b_head.log 2>&1
tail -1 gpurun_out/b_head.log | cut -c1-1800
