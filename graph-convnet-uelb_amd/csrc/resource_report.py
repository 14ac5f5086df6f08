#!/usr/bin/env python3
"""VGPR / AGPR / scratch / occupancy / LDS per kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
out = subprocess.run("/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage -o /tmp/_gcn_rr.so api.hip",
                     shell=True, capture_output=True, text=True, cwd=here).stderr
cur, rows = None, []
for line in out.split('\n'):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for r in rows:
    if re.search(pat, r["name"]):
        print("%-64s vgpr=%s agpr=%s scratch=%s occ=%s lds=%s" % (r["name"][:64], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))
