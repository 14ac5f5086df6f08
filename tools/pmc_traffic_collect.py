#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh (gpurun_out/pmc_traffic_<config>_<counter>/) into
gpurun_out/pmc_traffic.json: one entry per (workload, graphs per GPU, kernel) with the HBM bytes per launch, plus the
hash of the library sources the passes ran on.  Copy the file to profiles/pmc_traffic.json to have bench.py report it.
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE is taken as reported; both KiB."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))
from src.common.synthetic import CONFIGS  # noqa: E402
from src.gcn.hip import lib as L  # noqa: E402


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


entries = []
for cfg in sys.argv[1:]:
    cfg = cfg.split(":")[0]
    fetch = per_kernel(os.path.join(ROOT, "gpurun_out", f"pmc_traffic_{cfg}_FETCH_SIZE"), "FETCH_SIZE")
    write = per_kernel(os.path.join(ROOT, "gpurun_out", f"pmc_traffic_{cfg}_WRITE_SIZE"), "WRITE_SIZE")
    for k in sorted(fetch):
        if k not in write:
            continue
        name = k.split("(")[0].replace("void ", "")
        f_kb, n = fetch[k]
        w_kb, _ = write[k]
        entries.append({"workload": cfg, "graphs_per_gpu": CONFIGS[cfg]["B"], "kernel": name, "launches_averaged": n,
                        "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
                        "traffic_bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024))})
out = {"source_hash": L.source_hash(),
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over 'bench.py --config <cfg> --steps 2 "
                 "--warmup 1 --no-graph' (tools/pmc_traffic.sh), averaged over the launches of each kernel; FETCH_SIZE doubled "
                 "(gfx950 tallies 128-byte read requests at 64 bytes), WRITE_SIZE as reported; both in KiB",
       "entries": entries}
path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
for e in entries:
    print(e["workload"], e["kernel"], e["traffic_bytes_per_launch"] / 1e6, "MB per launch")
print("wrote", path)
