#!/usr/bin/env python3
"""Beam search at 200 nodes x 100 commodities (bench.beam_bench_large) on its own: one JSON line.  Under
rocprofv3 --kernel-trace --stats it gives the per-kernel times of the decode (python3 tools/beam_large_probe.py)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))
import torch
import bench
print(json.dumps(bench.beam_bench_large(torch.device("cuda:0"))))
