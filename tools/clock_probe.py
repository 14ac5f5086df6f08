#!/usr/bin/env python3
"""Which shader clock does the GPU hold while the training step replays back to back?  Runs the graph-replay step of
a configuration for a few seconds and samples the driver's current sclk level (sysfs pp_dpm_sclk, the line marked
with '*') from a second thread; prints the distribution.  Usage: python tools/clock_probe.py [config] [seconds]"""
import glob
import os
import sys
import threading
import time
from collections import Counter

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))


def sclk_files():
    return sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))


def current(path):
    try:
        for line in open(path):
            if "*" in line:
                return line.strip()
    except OSError as ex:
        return f"unreadable: {ex}"
    return "no active level"


def main():
    import bench
    from src.common.synthetic import CONFIGS, make_batch
    from src.gcn.hip.train_step import HipTrainStep
    cfg = sys.argv[1] if len(sys.argv) > 1 else "nsfnet_10_commodities"
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
    dev = torch.device("cuda:0")
    shape = dict(CONFIGS[cfg])
    batch = make_batch(shape["B"], shape["V"], shape["C"], shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"],
                       shape["dem_hi"], seed=1234)
    model = bench.build_model(shape, dev).train()
    step = HipTrainStep(model, shape["B"], dev, use_graph=True)
    step.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                    batch["edges_target"], batch["edge_cw"])
    files = sclk_files()
    print("sclk files:", files)
    for f in files[:8]:
        print(" idle", f, current(f))
    seen = Counter()
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            for f in files[:8]:
                seen[(f, current(f))] += 1
            time.sleep(0.02)

    th = threading.Thread(target=sampler)
    for _ in range(20):
        step.step()
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(200):
            step.step()
        torch.cuda.synchronize()
        n += 200
    el = time.time() - t0
    stop.set()
    th.join()
    print(f"{n} steps in {el:.2f} s = {el / n * 1e3:.4f} ms per step")
    for (f, lvl), k in sorted(seen.items()):
        print(f" busy {f} {lvl}: {k} samples")
    os.system("rocm-smi --showclocks 2>/dev/null | head -30")


if __name__ == "__main__":
    main()
