#!/usr/bin/env python3
"""Benchmark of the GCN hot path on MI355X: training graphs/sec (forward + backward).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (the driver launches N > 1 through torch.distributed.run); every rank runs the
workload on its own synthetic batch (weak scaling), gradients are averaged with one RCCL all-reduce of
the flat gradient buffer per step.  A "step" is one forward + backward of the full model (embedding,
L gated graph-conv layers, edge MLP, loss, all parameter gradients) over one batch that is already
resident in HBM; data loading and the optimizer are excluded, as in the metric's definition.
Rank 0 prints ONE JSON line.

Besides the headline value the line carries
  roofline     - the dominant kernel's algorithmic bytes per launch / its average duration, measured
                 with HIP events around each of its launches in an eager (non-graph) pass;
  cpu_baseline - the CPU oracle (oracle/gcn_oracle.py, a plain PyTorch restatement of the reference)
                 timed on this box's host cores on the same batch (train and eval-with-gradient mode, all
                 threads and one), rank 0, N = 1 only;
  batch_sweep  - the same step at B = 20, 80, 320 graphs (headline workload only);
  beam_search  - graphs/sec of the batched GPU beam-search decode next to the C oracle (N = 1 only), also at
                 200 nodes x 100 commodities with the bytes of scores read per second;
  large_configs - the two large benchmark configurations (100 nodes x 50 commodities at H = 300, L = 30 and 200 nodes x
                 100 commodities at H = 64, L = 10) at one graph per GPU: ms per step and the event-timed edge kernels.

Started as plain `python bench.py --gpus N` with N > 1 (no launcher, WORLD_SIZE unset) the script starts its N ranks
itself, one child process per GPU, before it touches the GPU, prints rank 0's JSON line and exits with the ranks' status.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "graph-convnet-uelb_amd"))

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak (spec)
COPY_RATE_GBS = 5460.0      # best streaming copy measured on this pool (not a peak: reported beside the roofline only)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA: 256 CUs x 256 flop per cycle x 2.4 GHz
DOMINANT_KERNEL = "k_edge_bwd"         # k_edge_bwd4 (H of 2 or 4 column blocks), k_edge_bwd3, k_edge_bwd_ws above 64


class Cfg(dict):
    __getattr__ = dict.__getitem__


def build_model(shape, device):
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    cfg = Cfg(num_nodes=shape["V"], num_commodities=shape["C"], voc_edges_in=3, voc_edges_out=2,
              hidden_dim=shape["H"], num_layers=shape["L"], mlp_layers=2, aggregation="mean",
              # dropout_rate: the key the reference's config files carry (nsfnet_10_commodities.json:38); the model reads
              # it and, like the reference, never uses it: the layers drop with p = 0.3 and the MLP with p = 0.2
              # (gcn_model.py:31,39,42), which is what the workload string reports
              dropout_rate=shape.get("dropout_rate", 0.3))
    torch.manual_seed(0)                      # identical initial weights on every rank
    return ResidualGatedGCNModel(cfg, torch.float32, torch.long).to(device)


def measured_traffic(config, graphs_per_gpu, kernel_prefix):
    """HBM bytes per launch of the dominant kernel from the committed counter passes (rocprofv3 --pmc cannot run inside
    this process): profiles/pmc_traffic.json holds one entry per (workload, graphs per GPU, kernel) together with the hash
    of the library sources the passes were taken on (tools/pmc_traffic.sh writes it).  Returns (bytes or None, stale):
    stale = an entry exists but was measured on other sources than the ones in the tree, and is therefore not reported."""
    from src.gcn.hip import lib as L
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None, False
    for e in table.get("entries", []):
        if e.get("workload") == config and e.get("graphs_per_gpu") == graphs_per_gpu and str(e.get("kernel", "")).startswith(kernel_prefix):
            if table.get("source_hash") != L.source_hash():
                return None, True
            return e.get("traffic_bytes_per_launch"), False
    return None, False


def live_traffic(config, graphs_per_gpu, layers, kernel_prefixes, timeout_s=150):
    """HBM bytes per launch of the named kernels, measured now: two child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: one counter per pass, counters only, no tracing), two eager steps
    each, on the same configuration.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes),
    WRITE_SIZE is taken as reported, both KiB.  Returns {prefix: bytes} for the kernels found, {} if a pass fails
    (no rocprofv3, a time limit, an unreadable result): the caller then falls back to the committed table."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    # not from inside a profiled run (tools/*.sh run this script under rocprofv3: no profiler inside a profiler)
    profiled = "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith("ROCPROF") for k in os.environ)
    if shutil.which("rocprofv3") is None or profiled:
        return {}
    sums = {}
    tmp = tempfile.mkdtemp(prefix="gcn_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for v in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = ["timeout", "-s", "KILL", str(timeout_s), "rocprofv3", "--pmc", ctr, "--kernel-include-regex", "k_edge_",
                   "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--config", config,
                   "--batch", str(graphs_per_gpu), "--layers", str(layers), "--steps", "2", "--warmup", "1", "--no-graph",
                   "--skip-cpu", "--skip-beam", "--skip-sweep", "--skip-large", "--skip-opt", "--skip-traffic"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            if r.returncode != 0:
                return {}
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == ctr:
                            acc.setdefault(row["Kernel_Name"].replace("void ", ""), []).append(float(row["Counter_Value"]))
            for pre in kernel_prefixes:
                vals = [x for k, v in acc.items() if k.startswith(pre) for x in v]
                # the wide backward edge kernel hands a ragged last column tile to k_edge_bwd_wp: one launch of each per
                # layer, their bytes belong together
                extra = [x for k, v in acc.items() if k.startswith("k_edge_bwd_wp") for x in v] if pre == "k_edge_bwd_ws" else []
                if not vals:
                    return {}
                kb = sum(vals) / len(vals) + (sum(extra) / len(extra) if extra else 0.0)
                sums[pre] = sums.get(pre, 0.0) + (2.0 * kb if ctr == "FETCH_SIZE" else kb) * 1024.0
    except Exception:
        return {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return {k: int(round(v)) for k, v in sums.items()}


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(shape, batch):
    """fwd+bwd of the CPU oracle on the same batch, in the two modes and at the two thread counts that matter:
    train mode (dropout on: what the trainer runs; more than half of the reference's CPU step is the dropout mask
    generator) and eval mode with gradients (dropout off: the parity mode), each with all host threads torch uses and
    with one thread, plus the train mode with autograd's anomaly detection on as the reference's trainer runs it.  A
    bounded sample: about 30 s of CPU work in all."""
    from oracle import gcn_oracle
    from src.gcn.models.gcn_model import ResidualGatedGCNModel
    cfg = Cfg(num_nodes=shape["V"], num_commodities=shape["C"], voc_edges_in=3, voc_edges_out=2,
              hidden_dim=shape["H"], num_layers=shape["L"], mlp_layers=2, aggregation="mean")
    torch.manual_seed(0)
    ref = ResidualGatedGCNModel(cfg, torch.float32, torch.long)
    P = {k: v.detach().clone().requires_grad_(True) for k, v in ref.state_dict().items()}
    ocfg = dict(num_layers=shape["L"], mlp_layers=2, aggregation="mean", voc_edges_out=2)
    cap = torch.from_numpy(batch["edges_capacity"])
    comm = torch.from_numpy(batch["commodities"])
    x_nodes = torch.from_numpy(batch["nodes"])
    y_edges = torch.from_numpy(batch["edges_target"])
    all_threads = torch.get_num_threads()

    def one(training):
        for p in P.values():
            p.grad = None
        _, loss = gcn_oracle.model_forward(P, ocfg, cap, comm, x_nodes, y_edges, batch["edge_cw"], training=training)
        loss.backward()

    def timed(training, threads, budget_s, max_steps, anomaly=False):
        torch.set_num_threads(threads)
        torch.autograd.set_detect_anomaly(anomaly)
        try:
            one(training)                             # warm-up
            times = []
            t_end = time.time() + budget_s
            while len(times) < max_steps and (time.time() < t_end or len(times) < 2):
                t0 = time.time()
                one(training)
                times.append(time.time() - t0)
        finally:
            torch.set_num_threads(all_threads)
            torch.autograd.set_detect_anomaly(False)
        med = float(np.median(times))
        return {"graphs_per_s": round(shape["B"] / med, 3), "ms_per_step": round(med * 1e3, 2), "threads": threads,
                "steps_timed": len(times)}

    modes = {
        "train_all_threads": timed(True, all_threads, 8.0, 40),
        "eval_grad_all_threads": timed(False, all_threads, 5.0, 40),
        # the reference's trainer leaves autograd's anomaly detection on for every step (src/gcn/train/trainer.py:100)
        "train_all_threads_anomaly_detection": timed(True, all_threads, 4.0, 20, anomaly=True),
        "train_1_thread": timed(True, 1, 4.0, 10),
        "eval_grad_1_thread": timed(False, 1, 4.0, 10),
    }
    head = modes["train_all_threads"]
    return {"value": head["graphs_per_s"], "unit": "graphs/s", "cores": int(all_threads), "kind": "port",
            "ms_per_step": head["ms_per_step"], "cpu_model": cpu_model_name(), "host_cpus": os.cpu_count(),
            "anomaly_detection": False,
            "sample": f"median of {head['steps_timed']} fwd+bwd steps of the same B={shape['B']} batch, train mode (dropout on), "
                      f"plain-PyTorch CPU oracle, torch threads={all_threads}; other modes and thread counts under 'modes' "
                      f"(median of the steps each ran in a 4-8 s budget, at least 2)",
            "modes": modes}


def batch_sweep(name, shape, device, rank, batches=(20, 80, 320), steps=20):
    """The headline step at larger batches of the same graphs (graph replay): 20 graphs are launch bound, this shows
    where the same kernels go with more rows per launch.  Whole-step algorithmic GB/s = (20 L + 20) H R_e bytes / time."""
    from src.common.synthetic import make_batch
    from src.gcn.hip.train_step import HipTrainStep
    out = []
    V, C, H, Lr = shape["V"], shape["C"], shape["H"], shape["L"]
    for B in batches:
        batch = make_batch(B, V, C, shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"], shape["dem_hi"],
                           seed=1234 + rank)
        model = build_model(shape, device).train()
        step = HipTrainStep(model, B, device, use_graph=True)
        step.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                        batch["edges_target"], batch["edge_cw"])
        for _ in range(5):
            step.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        step_bytes = (20.0 * Lr + 20.0) * H * (B * V * V * C)
        out.append({"graphs_per_gpu": B, "ms_per_step": round(ms, 4), "graphs_per_s": round(B / ms * 1e3, 1),
                    "whole_step_algorithmic_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1),
                    "whole_step_frac_of_hbm_peak": round(step_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
        del step, model
        torch.cuda.empty_cache()
    return {"workload": name, "mode": "train (dropout on), fwd+bwd, HIP graph replay", "points": out}


def time_kernels(eager, lib, prefixes, nprof):
    """Average launch duration (ms) of the kernels whose names start with each prefix, from the kernels' own begin / end
    time stamps on the launch stream (gcn_prof_begin / gcn_prof_end), over nprof eager steps per prefix."""
    out = {}
    for pre in prefixes:
        lib.gcn_prof_begin(pre.encode())
        for _ in range(nprof):
            eager.step()
        tot_ms, n = ctypes.c_double(0), ctypes.c_int(0)
        lib.gcn_prof_end(ctypes.byref(tot_ms), ctypes.byref(n))
        out[pre] = (tot_ms.value / max(n.value, 1), int(n.value))
    if "k_edge_bwd_ws" in out:
        # a ragged last column tile runs in k_edge_bwd_wp (one launch per layer beside k_edge_bwd_ws): one duration for both
        lib.gcn_prof_begin(b"k_edge_bwd_wp")
        for _ in range(nprof):
            eager.step()
        tot_ms, n = ctypes.c_double(0), ctypes.c_int(0)
        lib.gcn_prof_end(ctypes.byref(tot_ms), ctypes.byref(n))
        if n.value > 0:
            out["k_edge_bwd_ws"] = (out["k_edge_bwd_ws"][0] + tot_ms.value / n.value, out["k_edge_bwd_ws"][1])
    return out


def edge_kernel_names(plan, H):
    """(forward, backward) edge kernel of a launch plan: the family the library picks for this size."""
    if H > 64:
        return "k_edge_fwd_ws", "k_edge_bwd_ws"
    return ("k_edge_fwd4", "k_edge_bwd4") if plan["two_blocks"] else ("k_edge_fwd3", "k_edge_bwd3")


def edge_kernel_roofline(kernel, forward, avg_ms, nlaunch, B, V, C, H, two_blocks):
    """Roofline entry of one edge kernel launch (= one layer over the whole batch).  HBM bound up to 64 channels:
    algorithmic bytes 8 H (R_e + R_n) forward (read e, write e_tmp) and 12 H (R_e + R_n) backward (read de and the saved
    activation, write de_in); above 64 channels the H x H product per row outweighs the row's bytes (2 H^2 flop against
    12 H bytes: 50 flop per byte at H = 300, the balance point of fp32 MFMA against HBM is about 20): matrix-pipe bound,
    priced against the dense fp32 MFMA peak whatever arithmetic the kernel uses to reach fp32 accuracy."""
    Re, Rn = B * V * V * C, B * V * C
    algo_bytes = (8.0 if forward else 12.0) * H * (Re + Rn)
    gbps = algo_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(gbps / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": algo_bytes,
         "avg_launch_us": round(avg_ms * 1e3, 2), "launches_timed": nlaunch}
    if H <= 64:
        # what this kernel family itself must move per launch (the fused forms read more than the algorithmic minimum):
        # forward reads the previous layer's e_tmp and e_in and writes e_in and e_tmp (4 edge tensors); the one-block
        # backward reads de, e_tmp and the lower layer's e_tmp and writes de and det (5), the two-block backward reads
        # de, e_tmp, e_in and the lower layer's e_tmp and writes de (5)
        model = (16.0 if forward else 20.0) * H * Re
        r["kernel_model_bytes_per_launch"] = model
        r["kernel_model_GBps"] = round(model / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else 0.0
        r["kernel_family"] = "two column blocks per wavefront" if two_blocks else "one column block per wavefront"
    else:
        flops = 2.0 * H * H * Re
        tf = flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        r.update({"bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                  "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": flops,
                  "hbm_GBps": round(gbps, 1),
                  **({"duration_includes": "k_edge_bwd_wp (the launch that takes a ragged last column tile, when there is one)"}
                     if kernel == "k_edge_bwd_ws" else {}),
                  "arithmetic": "fp32 operands cut into three bf16 pieces, six v_mfma_f32_16x16x32_bf16 products per block, "
                                "fp32 accumulation (fp32-accurate; no fp32 MFMA issued)"})
    return r


def large_configs(device, rank):
    """The two large benchmark configurations at one graph per GPU: 2 warm-up + 3 timed steps of HIP graph replay, then
    the edge kernels timed by their own time stamps in an eager pass."""
    from src.common.synthetic import CONFIGS, make_batch
    from src.gcn.hip import lib as L
    from src.gcn.hip.train_step import HipTrainStep
    lib = L.load()
    out = {}
    for name in ("synthetic_100n_50c_h300_l30", "synthetic_200n_100c"):
        try:
            shape = dict(CONFIGS[name])
            B = 1
            V, C, H, Lr = shape["V"], shape["C"], shape["H"], shape["L"]
            batch = make_batch(B, V, C, shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"], shape["dem_hi"],
                               seed=1234 + rank)
            model = build_model(shape, device).train()
            args = (batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                    batch["edges_target"], batch["edge_cw"])
            step = HipTrainStep(model, B, device, use_graph=True)
            step.load_batch(*args)
            for _ in range(2):
                step.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                step.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            loss = float(step.loss_out[0].item())
            del step
            torch.cuda.empty_cache()
            eager = HipTrainStep(model, B, device, use_graph=False)
            eager.load_batch(*args)
            eager.step()
            plan = L.layer_plan(B, V, C, H)
            kf, kb = edge_kernel_names(plan, H)
            names = [kf, kb] + (["k_gemm_tn"] if H > 64 else [])
            tk = time_kernels(eager, lib, names, 1)
            step_bytes = (20.0 * Lr + 20.0) * H * (B * V * V * C)
            e = {"workload": f"V={V} C={C} H={H} L={Lr}, {B} graph per GPU, train mode, fwd+bwd, HIP graph replay",
                 "ms_per_step": round(ms, 3), "graphs_per_s": round(B / ms * 1e3, 3), "steps": 3, "warmup": 2,
                 "loss": round(loss, 5),
                 "whole_step_algorithmic_GBps": round(step_bytes / (ms * 1e-3) / 1e9, 1),
                 "edge_bwd": edge_kernel_roofline(kb, False, tk[kb][0], tk[kb][1], B, V, C, H, plan["two_blocks"]),
                 "edge_fwd": edge_kernel_roofline(kf, True, tk[kf][0], tk[kf][1], B, V, C, H, plan["two_blocks"])}
            if H > 64:
                ms_tn, n_tn = tk["k_gemm_tn"]
                flops_tn = 2.0 * H * H * (B * V * V * C + 3.0 * B * V * C)
                e["weight_gradient"] = {"kernel": "k_gemm_tn_ws (four products per launch: the edge product and three node products; "
                                                  "bf16 pieces, v_mfma_f32_32x32x16_bf16, fp32 accumulation)",
                                        "avg_launch_us": round(ms_tn * 1e3, 2), "launches_timed": n_tn,
                                        "achieved_TFLOPs_fp32_products": round(flops_tn / (ms_tn * 1e-3) / 1e12, 1)}
            out[name] = e
            del eager, model
            torch.cuda.empty_cache()
        except Exception as ex:      # the headline metric must still be reported
            out[name] = {"error": repr(ex)}
    return out


def beam_bench_large(device, B=2, V=200, C=100, K=10):
    """Beam-search decode at the largest benchmark configuration (BASELINE.json: 200 nodes x 100 commodities, K = 10, strict,
    deterministic): graphs/s, and the bytes of the score tensor y the search reads (4 V^2 bytes per commodity) over the
    decode time; the same decode by the C oracle on one host core next to it, paths compared bit for bit.  Scores favour
    each commodity's labelled path, as a trained model's would, so that graphs decode completely."""
    from oracle import beam_oracle_c
    from src.common.synthetic import make_batch
    from src.gcn.algorithms.beamsearch_uelb import BeamSearchFactory
    b = make_batch(B, V, C, "random", 1000, 5000, 5, 500, seed=300)
    rng = np.random.default_rng(V * 100 + C)
    p = rng.random((B, V, V, C)).astype(np.float32) * 0.3 + 0.01
    p[b["edges_target"] == 1] = 0.9
    y = np.log(p).astype(np.float32)
    cap = torch.from_numpy(b["edges_capacity"]).to(device)
    comm = torch.from_numpy(b["commodities"]).to(device)
    algo = BeamSearchFactory.create_algorithm("deterministic", y_pred_edges=torch.zeros((B, V, V, C), device=device), beam_size=K,
                                              batch_size=B, edges_capacity=cap, commodities=comm, dtypeFloat=torch.float32,
                                              dtypeLong=torch.long, mode_strict=True)
    algo.y = torch.from_numpy(y).to(device)
    algo.search()
    torch.cuda.synchronize()
    reps = 10
    t0 = time.time()
    for _ in range(reps):
        paths, plen, feas, att = algo._run()
    torch.cuda.synchronize()
    gpu_s = (time.time() - t0) / reps
    t0 = time.time()
    rp, rl, rf, _, _ = beam_oracle_c.search(y, b["edges_capacity"], b["commodities"], K, "deterministic", True)
    cpu_s = time.time() - t0
    same = bool(np.array_equal(paths.cpu().numpy(), rp) and np.array_equal(plen.cpu().numpy(), rl))
    y_bytes = 4.0 * B * V * V * C
    return {"workload": f"V={V} C={C} K={K} strict deterministic, {B} graphs", "graphs_per_s": round(B / gpu_s, 2),
            "ms_per_batch": round(gpu_s * 1e3, 3), "y_bytes": y_bytes, "y_GBps": round(y_bytes / gpu_s / 1e9, 2),
            "feasible_graphs": int(np.sum(rf)), "cpu_oracle_graphs_per_s": round(B / cpu_s, 2), "cpu_cores": 1,
            "paths_bit_exact_vs_oracle": same}


def beam_bench(shape, batch, device, K=10):
    """Deterministic strict beam search (the evaluator's setting, K = beam_size of the config) on the
    logits of the benchmark model: GPU kernel vs the C oracle on one host core."""
    from oracle import beam_oracle_c
    from src.gcn.algorithms.beamsearch_uelb import BeamSearchFactory
    B, V, C = shape["B"], shape["V"], shape["C"]
    g = torch.Generator().manual_seed(7)
    logits = (torch.randn((B, V, V, C, 2), generator=g) * 2.0).to(device)
    cap = torch.from_numpy(batch["edges_capacity"]).to(device)
    comm = torch.from_numpy(batch["commodities"]).to(device)
    algo = BeamSearchFactory.create_algorithm("deterministic", y_pred_edges=logits, beam_size=K, batch_size=B,
                                              edges_capacity=cap, commodities=comm, dtypeFloat=torch.float32,
                                              dtypeLong=torch.long, mode_strict=True)
    algo.search()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.time()
    for _ in range(reps):
        paths, plen, feas, att = algo._run()
    torch.cuda.synchronize()
    gpu_s = (time.time() - t0) / reps
    y = algo.y.cpu().numpy()
    t0 = time.time()
    rp, rl, rf, _, _ = beam_oracle_c.search(y, batch["edges_capacity"], batch["commodities"], K, "deterministic", True)
    cpu_s = time.time() - t0
    same = bool(np.array_equal(paths.cpu().numpy(), rp) and np.array_equal(plen.cpu().numpy(), rl))
    return {"algorithm": "deterministic", "beam_size": K, "mode_strict": True, "graphs_per_s": round(B / gpu_s, 1),
            "ms_per_batch": round(gpu_s * 1e3, 3), "cpu_oracle_graphs_per_s": round(B / cpu_s, 1), "cpu_cores": 1,
            "paths_bit_exact_vs_oracle": same}


def sampler_bench(shape, batch, device):
    """REINFORCE path sampling (temperature 1.5, top-p 0.95, eps 0.05: the reference's rl_finetuning config) on
    random edge scores of the benchmark shape: GPU sample + log-probability backward per batch vs the CPU
    restatement on a few graphs; the CPU paths must equal the GPU's for the same counter-based draws."""
    from oracle import sampler_oracle
    from src.gcn.algorithms.path_sampler import PathSampler
    B, V, C = shape["B"], shape["V"], shape["C"]
    g = torch.Generator().manual_seed(11)
    y = (torch.randn((B, V, V, C), generator=g) * 2.0)
    y[torch.from_numpy(batch["edges_capacity"] == 0).unsqueeze(-1).expand_as(y)] = -1e9
    yd = y.to(device).requires_grad_(True)
    cap = torch.from_numpy(batch["edges_capacity"]).to(device)
    comm = torch.from_numpy(batch["commodities"]).to(device)
    s = PathSampler(yd, cap, comm, temperature=1.5, top_p=0.95, entropy_epsilon=0.05)
    w = torch.ones(B, device=device)

    def one():
        yd.grad = None
        out = s._run()
        out[0].backward(w)
        return out

    one()
    torch.cuda.synchronize()
    reps = 20
    t0 = time.time()
    for _ in range(reps):
        out = one()
    torch.cuda.synchronize()
    gpu_s = (time.time() - t0) / reps
    seed, call = s.last_rng
    nb = min(B, 4)
    t0 = time.time()
    op, *_ = sampler_oracle.sample(y[:nb], batch["edges_capacity"][:nb], batch["commodities"][:nb], 1.5, 0.95, 0.05,
                                   uniform=sampler_oracle.counter_uniform(seed, call))
    cpu_s = (time.time() - t0) / nb
    paths, plen = out[1].cpu().numpy(), out[2].cpu().numpy()
    got = [[[int(v) for v in paths[b, c, :plen[b, c]]] for c in range(C)] for b in range(nb)]
    same = sum(got[b][c] == op[b][c] for b in range(nb) for c in range(C))
    return {"temperature": 1.5, "top_p": 0.95, "eps": 0.05, "graphs_per_s": round(B / gpu_s, 1),
            "ms_per_batch": round(gpu_s * 1e3, 3), "includes": "sampling, log-probabilities and their backward",
            "cpu_oracle_graphs_per_s": round(1.0 / cpu_s, 2), "cpu_cores": 1,
            "paths_equal_to_oracle": f"{same}/{nb * C}"}


def _stage(msg):
    """Progress line on stderr (GCN_BENCH_VERBOSE=1): which rank reached which stage."""
    if os.environ.get("GCN_BENCH_VERBOSE"):
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: one child process per GPU, started before this process makes any
    GPU call (importing torch does not initialise the device).  Rank 0's stdout is passed through; the exit status is
    the first non-zero status of a rank (the others are then stopped), 0 if all ranks succeed."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    status = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            rc = p.poll()
            if rc is None:
                continue
            live.remove(p)
            if rc != 0 and status == 0:
                status = rc
                for q in live:          # a rank failed: the others would wait in a collective forever
                    q.terminate()
    return status if status >= 0 else 128 - status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="nsfnet_10_commodities")
    ap.add_argument("--batch", type=int, default=0, help="graphs per GPU (default: the config's batch_size)")
    ap.add_argument("--layers", type=int, default=0, help="conv layers (default: the config's num_layers; tests use fewer)")
    ap.add_argument("--no-graph", action="store_true", help="launch kernels eagerly instead of replaying a HIP graph")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-beam", action="store_true")
    ap.add_argument("--skip-sweep", action="store_true", help="skip the batch sweep (B = 20, 80, 320) of the headline workload")
    ap.add_argument("--skip-opt", action="store_true", help="skip the extra leg that adds the optimizer step")
    ap.add_argument("--bucket-mb", type=float, default=8.0,
                    help="with a process group: gradients of at least this many MB are averaged in buckets of this size that "
                         "start while the backward pass still runs (eager launches); below it, one message after the step")
    ap.add_argument("--skip-large", action="store_true", help="skip the large_configs legs (the two large configurations at one graph per GPU)")
    ap.add_argument("--skip-traffic", action="store_true",
                    help="do not start the two rocprofv3 counter passes that measure the edge kernels' HBM traffic")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run the broadcast / all-reduce even with one rank "
                         "(executes the RCCL path on a single GPU)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))

    from src.common.synthetic import CONFIGS, make_batch
    from src.gcn.hip import lib as L
    from src.gcn.hip.train_step import HipTrainStep

    from src.gcn.hip import dist as D
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start the ranks with `python bench.py --gpus N` or "
                         f"`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    dist = None
    backend = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # GCN_BENCH_BACKEND=gloo rehearses the multi-process path on a box with fewer GPUs than ranks (ranks then
        # share devices round-robin); the measured configuration is always nccl (= RCCL) with one GPU per rank
        backend = os.environ.get("GCN_BENCH_BACKEND", "nccl")
        ndev = max(torch.cuda.device_count(), 1)
        if backend == "nccl":
            if local_rank >= ndev:
                raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({ndev} visible)")
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            local_rank = local_rank % ndev
            dist.init_process_group(backend, rank=rank, world_size=world)
        D.RUN_SINGLE_RANK_COLLECTIVES = bool(args.force_dist)     # one rank: still broadcast and all-reduce
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    _stage(f"process group ready, device {device}")

    shape = dict(CONFIGS[args.config])
    if args.batch:
        shape["B"] = args.batch
    if args.layers:
        shape["L"] = args.layers
    B, V, C, H, Lr = shape["B"], shape["V"], shape["C"], shape["H"], shape["L"]
    batch = make_batch(B, V, C, shape["topology"], shape["cap_lo"], shape["cap_hi"], shape["dem_lo"], shape["dem_hi"],
                       seed=1234 + rank)
    model = build_model(shape, device).train()          # train mode: dropout active, as the trainer runs
    if dist is not None:
        D.broadcast_parameters(model.parameters())
        # class weights: one set for the whole job (rank 0's), as the trainer uses global weights for every shard
        cw = torch.from_numpy(np.asarray(batch["edge_cw"], dtype=np.float32)).to(device)
        D.broadcast_(cw)
        batch["edge_cw"] = cw.cpu().numpy()
    _stage("parameters broadcast")
    # Large gradients (the H = 300, L = 30 configuration: 44 MB) are averaged in buckets behind per-layer completion
    # events of the backward pass, which needs eager launches (milliseconds per kernel there: launch cost is nothing);
    # small ones (0.7 MB at the headline) in one message after the replayed graph.
    nparam_bytes = 4 * sum(p.numel() for p in model.parameters())
    bucketed = dist is not None and nparam_bytes >= args.bucket_mb * 2 ** 20
    use_graph = not args.no_graph and not bucketed
    step = HipTrainStep(model, B, device, use_graph=use_graph)
    step.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                    batch["edges_target"], batch["edge_cw"])
    buckets = D.BucketedGradAllreduce(step, bucket_bytes=int(args.bucket_mb * 2 ** 20),
                                      min_total_bytes=int(args.bucket_mb * 2 ** 20)) if bucketed else None

    def one_step():
        step.step()
        if buckets is not None:
            buckets.after_step()                    # per-bucket messages on a side stream, behind the layers' events
        elif dist is not None:
            D.allreduce_mean_(step.flat_grad)       # one message; on RCCL the 1/N is part of the collective (AVG)

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    _stage("warm-up done")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        D.allreduce_max_(t)
        elapsed = float(t.item())
    _stage("timed steps done")
    loss = float(step.loss_out[0].item())           # of the last timed step
    # ---- the gradient all-reduce on its own: time between the end of the step's last kernel and the end of the
    # collective, over steps of their own (outside the headline's timed region)
    allreduce_ms = None
    if dist is not None:
        n_ar = max(3, min(args.steps, 20))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_ar)]
        wall = 0.0
        for e0, e1 in ev:
            step.step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e0.record()
            D.allreduce_mean_(step.flat_grad)
            e1.record()
            torch.cuda.synchronize()
            wall += time.perf_counter() - t1
        t = torch.tensor([sum(a.elapsed_time(b) for a, b in ev) / n_ar, wall / n_ar * 1e3], dtype=torch.float64, device=device)
        D.allreduce_max_(t)
        allreduce_ms = [float(v) for v in t.tolist()]
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    # ---- the same steps followed by the optimizer (multi-tensor Adam of the library), reported beside the headline:
    # the metric itself is forward + backward (+ gradient all-reduce), as the baseline names it
    with_opt = None
    if not args.skip_opt:
        from src.gcn.hip.fused_adam import FusedAdam
        opt = FusedAdam(model.parameters(), lr=1e-4)

        def full_step():
            one_step()
            opt.step()

        for _ in range(max(1, min(args.warmup, 5))):
            full_step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full_step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            D.allreduce_max_(t)
            el = float(t.item())
        with_opt = {"optimizer": "Adam (gcn_adam_step, one launch for all parameters)",
                    "ms_per_step": round(el / args.steps * 1e3, 4), "graphs_per_s": round(world * B * args.steps / el, 1)}
        _stage("steps with optimizer done")

    # ---- the edge kernels, timed by their own begin / end time stamps on the launch stream (eager pass, rank 0);
    # `roofline` is the dominant one (edge backward), `roofline_fwd` the forward kernel of the same layer
    roofline = roofline_fwd = None
    if rank == 0:
        lib = L.load()
        eager = HipTrainStep(model, B, device, use_graph=False)
        eager.load_batch(batch["edges_capacity"], batch["commodities"][:, :, 2].astype(np.float32), batch["nodes"],
                         batch["edges_target"], batch["edge_cw"])
        for _ in range(3):
            eager.step()
        torch.cuda.synchronize()
        plan = L.layer_plan(B, V, C, H)
        kf, kb = edge_kernel_names(plan, H)
        nprof = max(5, min(args.steps, 20)) if H <= 64 else 2
        tk = time_kernels(eager, lib, [kb, kf], nprof)
        roofline = edge_kernel_roofline(kb, False, tk[kb][0], tk[kb][1], B, V, C, H, plan["two_blocks"])
        roofline_fwd = edge_kernel_roofline(kf, True, tk[kf][0], tk[kf][1], B, V, C, H, plan["two_blocks"])
        step_bytes = (20.0 * Lr + 20.0) * H * (B * V * V * C)
        roofline["whole_step_algorithmic_GBps"] = round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1)
        for p, gview in zip(step.params, step.grad_views):   # the eager helper re-pointed p.grad; restore
            p.grad = gview
        del eager
        # HBM traffic per launch: measured by counter passes this run starts itself; the committed table of earlier
        # passes (profiles/pmc_traffic.json, only if taken on these sources) stands in where that is not possible.
        # The child runs need the device memory this process holds for its own steps (the large configurations keep
        # tens of GB of saved activations per step object): release it first.
        grad_bytes = int(step.flat_grad.numel()) * 4
        live = {}
        if not args.skip_traffic and world == 1:
            if dist is None and buckets is None:
                for p in step.params:
                    p.grad = None
                step = None                     # (one_step and full_step see the same variable: nothing else holds the object)
                torch.cuda.empty_cache()
            live = live_traffic(args.config, B, Lr, [kb, kf])
        for rf, name in ((roofline, kb), (roofline_fwd, kf)):
            if name in live:
                rf["traffic"] = live[name]
                rf["traffic_stale"] = False
                rf["traffic_source"] = ("two rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE) started by this run, "
                                        "two eager steps each")
            else:
                rf["traffic"], rf["traffic_stale"] = measured_traffic(args.config, B, name)
                rf["traffic_source"] = "profiles/pmc_traffic.json (committed counter passes)"
            if rf["traffic"]:
                rf["traffic_over_algorithmic"] = round(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 3)
                if rf.get("avg_launch_us"):
                    # the rate the kernel moves its real bytes at, next to what a plain streaming copy reaches on this
                    # pool's boxes (tools/copy_peak_bench.hip: 5.46 TB/s, profiles/r03_copy_bandwidth.txt)
                    rf["traffic_GBps"] = round(rf["traffic"] / (rf["avg_launch_us"] * 1e-6) / 1e9, 1)
                    rf["traffic_frac_of_measured_copy_rate"] = round(rf["traffic_GBps"] / COPY_RATE_GBS, 3)

    out = None
    if rank == 0:
        out = {
            "metric": "GCN train graphs/sec (fwd+bwd)", "value": round(value, 1), "unit": "graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs/gcn/{args.config}.json: V={V} C={C} H={H} L={Lr} mlp_layers=2 mean "
                                   f"aggregation, {B} graphs per GPU, train mode (dropout 0.3/0.2 on), fwd+bwd, "
                                   f"{'HIP graph replay' if use_graph else 'eager launches'}",
                       "global_batch": B * world, "parallelism": f"dp{world}"},
            "loss": round(loss, 5),
            # ranks of the RCCL (backend "nccl") process group the gradient all-reduce ran on; 0 = no process group
            "rccl_ranks": D.rccl_ranks(),
            "gradient_allreduce": None if dist is None else {
                "backend": backend, "op": "all_reduce AVG" if backend == "nccl" else "all_reduce SUM, host-staged",
                "bytes": grad_bytes, "per_step": 1,
                # max over ranks: device time between events recorded around the collective on the step's stream, and
                # host wall time of the same stretch (collective launch to completion), step already finished
                "ms_per_step": round(allreduce_ms[0], 4), "wall_ms_per_step": round(allreduce_ms[1], 4),
                "overlap": ("none (one message after the step's last kernel)" if buckets is None else
                            f"{len(buckets.buckets)} buckets of >= {args.bucket_mb:g} MB on a side stream, each behind the completion "
                            f"event of its last layer (gcn_model_bwd_ev)"),
                "buckets": None if buckets is None else [[int(lo) * 4, int(hi) * 4] for lo, hi, _ in buckets.buckets]},
            "roofline": roofline, "roofline_fwd": roofline_fwd, "with_optimizer": with_opt,
        }
        if world == 1 and not args.skip_cpu:
            out["cpu_baseline"] = cpu_baseline(shape, batch)
        if world == 1 and not args.skip_sweep and args.config == "nsfnet_10_commodities" and not args.batch:
            try:
                out["batch_sweep"] = batch_sweep(args.config, shape, device, rank)
            except Exception as ex:      # the headline metric must still be reported
                out["batch_sweep"] = {"error": repr(ex)}
        if world == 1 and not args.skip_large and args.config == "nsfnet_10_commodities" and not args.batch:
            step = None
            torch.cuda.empty_cache()
            out["large_configs"] = large_configs(device, rank)
        if world == 1 and not args.skip_beam:
            try:
                out["beam_search_200n_100c"] = beam_bench_large(device)
            except Exception as ex:
                out["beam_search_200n_100c"] = {"error": repr(ex)}
        if world == 1 and not args.skip_beam:
            try:
                out["beam_search"] = beam_bench(shape, batch, device)
            except Exception as ex:      # the headline metric must still be reported
                out["beam_search"] = {"error": repr(ex)}
            if shape["V"] <= 256:
                try:
                    out["path_sampler"] = sampler_bench(shape, batch, device)
                except Exception as ex:
                    out["path_sampler"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
