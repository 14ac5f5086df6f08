#!/bin/bash
# Prints VGPR / AGPR / scratch / occupancy per kernel of the library (hipcc -Rpass-analysis).
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Rpass-analysis=kernel-resource-usage \
   -o /tmp/_gcn_rr.so api.hip 2>&1 | python3 -c '
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    for key in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","VGPRs Spill","LDS Size \[bytes/block\]"):
        m=re.search(r"remark: [^ ]+ +"+key+r": (\d+)",line)
        if m and cur is not None: cur[key]=m.group(1)
for r in rows:
    print("%-70s vgpr=%s agpr=%s scratch=%s occ=%s" % (r["name"][:70], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize \\[bytes/lane\\]"), r.get("Occupancy \\[waves/SIMD\\]")))
'
