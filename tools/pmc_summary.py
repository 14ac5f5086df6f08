#!/usr/bin/env python3
"""Average each counter per kernel name over the dispatches of a rocprofv3 --pmc pass."""
import csv, glob, sys, collections
d = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("   %-32s n=%4d avg=%14.1f" % (c, len(v), sum(v) / len(v)))
