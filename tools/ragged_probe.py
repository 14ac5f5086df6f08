#!/usr/bin/env python3
"""How much does a ragged last column tile cost the wide backward edge kernel?  (measurement aid)

Runs the bench step of the H = 300 configuration with V = 96 (six full tiles of 16 columns j), V = 100 (six tiles and one
with 4 of 16 columns in use) and V = 112 (seven full tiles) at 4 layers and prints the edge kernels' time per launch and
per million edge rows.  Usage (GPU box): python tools/ragged_probe.py"""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))
import bench  # noqa: E402
from src.common.synthetic import CONFIGS  # noqa: E402

base = CONFIGS["synthetic_100n_50c_h300_l30"]
for V in (96, 100, 112):
    name = f"probe_v{V}"
    CONFIGS[name] = dict(base, V=V)
    sys.argv = ["bench.py", "--config", name, "--layers", "4", "--steps", "3", "--warmup", "2", "--skip-cpu", "--skip-beam",
                "--skip-sweep", "--skip-large", "--skip-traffic", "--skip-opt"]
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.main()
    d = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
    rows = base["B"] * V * V * base["C"] / 1e6
    b, f = d["roofline"]["avg_launch_us"], d["roofline_fwd"]["avg_launch_us"]
    print(f"V = {V}: step {d['ms_per_step']:.2f} ms; {rows:.3f} M edge rows; backward {b:.0f} us = {b / rows:.0f} us per M rows; forward {f:.0f} us = {f / rows:.0f} us per M rows", flush=True)
