#!/usr/bin/env python3
"""Phase timing of one wavefront of the forward edge kernel (measurement aid).

Needs a library built with -DGCN_PHASE_TRACE (hipcc ... -DGCN_PHASE_TRACE -o lib/libgcn_trace.so csrc/api.hip) and
GCN_UELB_LIB pointing at it.  Runs a few eager training steps at nsfnet_10 and prints, per step of workgroup 0 /
wavefront 0 in the last k_edge_fwd3 launch, the shader-clock cycles spent in: phase A (input tile to LDS), coordinate
bookkeeping + load issue, the barrier, LDS reads + MFMA, waiting for the epilogue rows, epilogue."""
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

cfg_name = sys.argv[1] if len(sys.argv) > 1 else "nsfnet_10_commodities"
shape = CONFIGS[cfg_name]
dev = torch.device("cuda:0")
model = build_model(shape, dev)
b = make_batch(shape["B"], shape["V"], shape["C"], shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"], shape["dem_hi"], seed=1234)
step = HipTrainStep(model, shape["B"], dev, use_graph=False)
which = sys.argv[2] if len(sys.argv) > 2 else "fwd"      # wide configurations: build with -DGCN_TRACE_BWD for the backward kernel
step.load_batch(b["edges_capacity"], b["commodities"][:, :, 2].astype(np.float32), b["nodes"], b["edges_target"], b["edge_cw"])
for _ in range(2 if shape["H"] > 64 else 5):
    step.step()
torch.cuda.synchronize()
lib = L.load()
buf = (ctypes.c_ulonglong * 512)()
lib.gcn_phase_trace_read.argtypes = [ctypes.c_void_p]
assert lib.gcn_phase_trace_read(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(64, 8)
names = (["nodeside", "phaseA", "barrier", "product0", "blocks..", "barrier2", "next"] if which == "bwd" else
         ["phaseA", "barrier", "product0", "epilogue0", "blocks1..", "barrier2", "next"] if shape["H"] > 64 else
         ["phaseA", "issue", "barrier", "lds+mfma", "epi-wait", "epilogue", "loop"])
print("step " + " ".join(f"{n:>9}" for n in names))
for s in range(16):
    if t[s, 0] == 0:
        break
    d = [t[s, k + 1] - t[s, k] for k in range(6)]
    nxt = t[s + 1, 0] - t[s, 6] if s + 1 < 16 and t[s + 1, 0] else 0
    print(f"{s:4d} " + " ".join(f"{int(v):9d}" for v in d + [nxt]), " total", int(t[s, 6] - t[s, 0]))
