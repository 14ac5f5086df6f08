#!/usr/bin/env python3
"""Print a per-kernel table from a rocprofv3 --stats csv directory: tools/prof_summary.py gpurun_out/prof_<tag> [steps]"""
import csv, glob, sys
d = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 48
f = glob.glob(d + '/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"total GPU ms {tot/1e6:.2f}  per step ms {tot/1e6/steps:.3f}")
for r in rows[:24]:
    print(f"{r['Name'][:46]:46s} calls/step={int(r['Calls'])/steps:6.1f} avg_us={float(r['AverageNs'])/1e3:8.2f} us/step={float(r['TotalDurationNs'])/1e3/steps:8.1f} pct={float(r['Percentage']):5.1f}")
