#!/usr/bin/env python3
"""How much does a ragged last column tile cost the strip edge kernels (H = 64)?  (measurement aid)

The 200-node configuration at 2 graphs with V = 192 (12 full tiles), 200 (12 full tiles and one with 8 of 16 columns) and
208 (13 full tiles); and the default2 shape with V = 32, 40, 48.  Usage (GPU box): python tools/ragged_probe_strip.py"""
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

for cfg, vs, batch in (("synthetic_200n_100c", (192, 200, 208), 2), ("default2", (32, 40, 48), 0)):
    base = CONFIGS[cfg]
    for V in vs:
        name = f"probe_{cfg}_v{V}"
        CONFIGS[name] = dict(base, V=V)
        sys.argv = ["bench.py", "--config", name, "--layers", "4", "--steps", "5", "--warmup", "2", "--skip-cpu", "--skip-beam",
                    "--skip-sweep", "--skip-large", "--skip-traffic", "--skip-opt"] + (["--batch", str(batch)] if batch else [])
        buf = io.StringIO()
        with redirect_stdout(buf):
            bench.main()
        d = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][-1])
        B = batch or base["B"]
        rows = B * V * V * base["C"] / 1e6
        b, f = d["roofline"]["avg_launch_us"], d["roofline_fwd"]["avg_launch_us"]
        print(f"{cfg} V = {V}: {rows:.3f} M edge rows; backward {b:.1f} us = {b / rows:.1f} us per M rows; forward {f:.1f} us = {f / rows:.1f} us per M rows", flush=True)
