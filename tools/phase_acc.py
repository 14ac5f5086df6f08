#!/usr/bin/env python3
"""Where the wavefronts of a wide edge kernel spend their cycles, summed over a whole launch (measurement aid).

Needs a library built with -DGCN_PHASE_ACC=1 (forward edge kernel) or =2 (backward) and GCN_UELB_LIB pointing at it.
Usage: GCN_UELB_LIB=... python tools/phase_acc.py <config> fwd|bwd"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))
sys.path.insert(0, ROOT)
from bench import build_model  # noqa: E402
from src.common.synthetic import CONFIGS, make_batch  # noqa: E402
from src.gcn.hip import lib as L  # noqa: E402
from src.gcn.hip.train_step import HipTrainStep  # noqa: E402

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_100n_50c_h300_l30"
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"
shape = CONFIGS[cfg_name]
dev = torch.device("cuda:0")
model = build_model(shape, dev)
b = make_batch(shape["B"], shape["V"], shape["C"], shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"], shape["dem_hi"], seed=1234)
step = HipTrainStep(model, shape["B"], dev, use_graph=False)
step.load_batch(b["edges_capacity"], b["commodities"][:, :, 2].astype(np.float32), b["nodes"], b["edges_target"], b["edge_cw"])
for _ in range(2):
    step.step()
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * (1024 * 12 * 8))()
lib.gcn_phase_acc_read.argtypes = [ctypes.c_void_p]
assert lib.gcn_phase_acc_read(buf) == 0
t = np.array(buf[:], dtype=np.float64).reshape(1024, 12, 8)
used = t.sum(axis=(1, 2)) > 0
t = t[used]
# forward kernel: wavefronts 0..7 multiply (slots: wait for the planes, product, epilogue, loop end, wait for the stagers),
# wavefronts 8..11 stage (slots: cut into planes, wait, fetch + form the next step's rows, wait for the product)
names = (["cut / -", "wait planes", "product / fetch", "epilogue", "loop end", "wait other role", "tail", "-"] if which == "fwd" else
         ["node side", "barrier", "stage", "barrier", "product", "epilogue", "loop end", "barrier3"])
tot = t.sum(axis=2)
print(f"{int(used.sum())} workgroups; cycles per wavefront: mean {tot.mean():.0f}, min {tot.min():.0f}, max {tot.max():.0f}")
for w in range(12):
    if tot[:, w].sum() == 0:
        continue
    print(f"wavefront {w}: " + "  ".join(f"{n} {t[:, w, k].mean() / tot[:, w].mean() * 100:5.1f}%" for k, n in enumerate(names) if n != "-"))
print("all       : " + "  ".join(f"{n} {t[:, :, k].sum() / tot.sum() * 100:5.1f}%" for k, n in enumerate(names) if n != "-"))
