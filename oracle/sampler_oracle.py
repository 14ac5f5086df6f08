"""CPU restatement of the reference's PathSampler (TEST INFRASTRUCTURE: imported by tests/ only, never
by the product path).

Follows src/gcn/algorithms/path_sampler.py of the reference step by step, in torch on the CPU so that
autograd yields the gradient of the log-probabilities the way it does there.  Two things are pluggable,
because the reference draws from torch's global generator and a GPU cannot replay that stream:

* ``uniform(bc, step)`` supplies one number in [0, 1) per step; the nucleus draw and the uniform fallback
  are the inverse-CDF picks that number selects (same distribution as torch.multinomial / torch.randint);
  ``counter_uniform`` is the hash the HIP kernel uses, so seeded runs can be compared path by path;
* ``forced_paths`` replays given paths (e.g. the ones the reference sampled) and returns their
  log-probabilities and entropies, which are deterministic functions of the path.

Pinned by tests/golden/sampler_*.npz, produced by running the reference itself (oracle/gen_golden.py).
"""
from __future__ import annotations

from collections import deque

import numpy as np
import torch

M32 = 0xFFFFFFFF


def _mix(x):
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


def counter_uniform(seed, call):
    """The kernel's per-step uniform: 24 bits of a hash of (seed, call, graph * C + commodity, step)."""
    h0 = _mix((seed & M32) ^ 0x9E3779B9)
    h0 = _mix(h0 ^ ((seed >> 32) & M32))
    h0 = _mix(h0 ^ (call & M32))
    h0 = _mix(h0 ^ ((call >> 32) & M32) ^ 0x85EBCA6B)

    def u(bc, step):
        h = _mix((h0 + bc * 0x9E3779B9) & M32)
        h = _mix(h ^ ((step * 0xC2B2AE35 + 0x165667B1) & M32))
        return np.float32(h >> 8) * np.float32(1.0 / 16777216.0)

    return u


def reachability(cap):
    """reach[b, i, j]: j reachable from i over edges with cap > 0 (path_sampler.py:220-260)."""
    cap = np.asarray(cap)
    B, V, _ = cap.shape
    reach = np.zeros((B, V, V), dtype=bool)
    for b in range(B):
        nbrs = [[j for j in range(V) if j != i and cap[b, i, j] > 0] for i in range(V)]
        for s in range(V):
            reach[b, s, s] = True
            todo = deque([s])
            while todo:
                n = todo.popleft()
                for m in nbrs[n]:
                    if not reach[b, s, m]:
                        reach[b, s, m] = True
                        todo.append(m)
    return reach


def nucleus(probs, top_p):
    """Indices (descending probability, ties by lower index) and renormalised masses of the top-p nucleus
    (path_sampler.py:409-423).  Returns None if every probability is <= 1e-8."""
    p = probs * (probs > 1e-8)
    if float(p.sum()) == 0.0:
        return None
    p = p / p.sum()
    order = torch.sort(p, descending=True, stable=True).indices
    sp = p[order]
    cum = torch.cumsum(sp, 0)
    over = torch.where(cum > top_p)[0]
    n = int(over[0]) + 1 if len(over) else len(sp)
    keep = sp[:n]
    sel = keep > 0
    return order[:n][sel], (keep / keep.sum())[sel]


def sample(y, cap, comm, temperature=1.0, top_p=0.8, eps=0.0, uniform=None, forced_paths=None, margin_out=None):
    """Returns (paths[b][c], logp [B] tensor attached to y, is_feasible, entropies[b][c][step], logp_bc [B,C]).

    y: [B,V,V,C] tensor (may require grad), cap [B,V,V], comm [B,C,3] integer array-like.
    margin_out: optional list; receives (b, c, step, distance of the draw to the nearest CDF boundary).
    """
    y = torch.as_tensor(y)
    cap_t = torch.as_tensor(np.asarray(cap), dtype=y.dtype)
    comm = np.asarray(comm)
    B, V, _, C = y.shape
    reach = torch.from_numpy(reachability(np.asarray(cap)))
    eye = torch.eye(V, dtype=torch.bool)
    paths, ents, logp_rows = [], [], []
    for b in range(B):
        invalid = (cap_t[b] <= 0) | eye                                           # :86
        logits = y[b].masked_fill(invalid.unsqueeze(-1), -1e20)                    # :106
        probs_all = torch.softmax(logits / temperature, dim=1) * (~invalid).unsqueeze(-1).to(y.dtype)   # :111-114
        g_paths, g_ents, g_logp = [], [], []
        for c in range(C):
            src, dst = int(comm[b, c, 0]), int(comm[b, c, 1])
            bc = b * C + c
            given = None if forced_paths is None else [int(v) for v in forced_paths[b][c]]
            p = probs_all[:, :, c]
            path, lp, step_ents, cur = [src], torch.zeros((), dtype=y.dtype), [], src
            for step in range(2 * V):                                             # :292-296
                if cur == dst:
                    break
                free = torch.ones(V, dtype=torch.bool)
                free[path] = False
                allowed = free & reach[b, :, dst]                                 # :305-315
                out = torch.clamp(p[cur] * allowed.to(y.dtype), min=0.0)
                total = out.sum()
                support = (cap_t[b, cur] > 0) & allowed
                u = None if uniform is None else float(uniform(bc, step))
                if float(total.detach()) <= 1e-20:                                         # :326-372
                    idx = torch.where(support)[0]
                    if len(idx) == 0:
                        lp = lp + torch.tensor(np.log(1e-8), dtype=y.dtype)
                        break
                    cand = p[cur][idx]
                    assert float(cand.detach().sum()) <= 1e-10, "model-guided fallback branch is unreachable"
                    if given is not None:
                        if len(path) >= len(given):
                            break
                        nxt = given[len(path)]
                    else:
                        nxt = int(idx[min(int(np.float32(u) * np.float32(len(idx))), len(idx) - 1)])
                    lp = lp + torch.tensor(np.log(1.0 / len(idx)), dtype=y.dtype)
                    path.append(nxt)
                    cur = nxt
                    continue
                if given is not None and len(path) >= len(given):
                    break
                out = out / total                                                 # :374
                if eps > 0.0 and bool(support.any()):                             # :376-391
                    uni = support.to(y.dtype)
                    uni = uni / (uni.sum() + 1e-8)
                    out = (1.0 - eps) * out + eps * uni
                    out = out / (out.sum() + 1e-8)
                step_ents.append(float(-(out * torch.log(out + 1e-8)).sum().detach()))      # :394-395
                if given is not None:
                    nxt = given[len(path)]
                else:
                    nuc = nucleus(out.detach(), top_p)
                    if nuc is None:
                        nxt = min(int(np.float32(u) * np.float32(V)), V - 1)
                    else:
                        idx, mass = nuc
                        cdf = torch.cumsum(mass, 0).numpy()
                        k = min(int(np.searchsorted(cdf, u, side="right")), len(idx) - 1)
                        nxt = int(idx[k])
                        if margin_out is not None:
                            edges = np.concatenate([[0.0], cdf])
                            margin_out.append((b, c, step, float(np.min(np.abs(edges - u)))))
                lp = lp + torch.log(out[nxt] + 1e-8)                              # :400
                path.append(nxt)
                cur = nxt
            g_paths.append(path)
            g_ents.append(step_ents)
            g_logp.append(lp)
        paths.append(g_paths)
        ents.append(g_ents)
        logp_rows.append(torch.stack(g_logp))
    logp_bc = torch.stack(logp_rows)
    feasible = True                                                               # :428-446
    for b in range(B):
        use = np.zeros((V, V))
        for c in range(C):
            for s, t in zip(paths[b][c][:-1], paths[b][c][1:]):
                use[s, t] += float(comm[b, c, 2])
        if (use > np.asarray(cap)[b] + 1e-6).any():
            feasible = False
            break
    return paths, logp_bc.sum(1), feasible, ents, logp_bc


def first_hop_distribution(y, cap, comm, b, c, temperature, top_p, eps):
    """Exact distribution of the first node after the source under the sampler (for frequency tests)."""
    y = torch.as_tensor(y)
    V = y.shape[1]
    cap_t = torch.as_tensor(np.asarray(cap), dtype=y.dtype)
    src, dst = int(comm[b][c][0]), int(comm[b][c][1])
    reach = torch.from_numpy(reachability(np.asarray(cap)))
    invalid = (cap_t[b] <= 0) | torch.eye(V, dtype=torch.bool)
    logits = y[b].masked_fill(invalid.unsqueeze(-1), -1e20)
    p = (torch.softmax(logits / temperature, dim=1) * (~invalid).unsqueeze(-1).to(y.dtype))[src, :, c]
    free = torch.ones(V, dtype=torch.bool)
    free[src] = False
    allowed = free & reach[b, :, dst]
    out = p * allowed.to(y.dtype)
    out = out / out.sum()
    support = (cap_t[b, src] > 0) & allowed
    if eps > 0.0 and bool(support.any()):
        uni = support.to(y.dtype)
        uni = uni / (uni.sum() + 1e-8)
        out = (1.0 - eps) * out + eps * uni
        out = out / (out.sum() + 1e-8)
    idx, mass = nucleus(out, top_p)
    dist = np.zeros(V)
    dist[idx.numpy()] = mass.numpy()
    return dist
