"""gcn_load_factor_static (usage scratch in a fixed array of the library's code object, no allocation) against
gcn_load_factor (usage scratch passed by the caller), through the C ABI: two calls in a row, the second with a
smaller input, so that the second call meets the first call's counts in the shared array.  Integer usage
counts and a float maximum: results must be identical bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a ROCm GPU")
    return torch.device("cuda:0")


def _random_paths(rng, B, V, C, PL):
    """Random simple paths from each commodity's source; about a fifth stop early (incomplete), one graph uses a
    zero-capacity edge (load factor +inf)."""
    cap = rng.integers(1000, 5000, size=(B, V, V)).astype(np.float32)
    cap[0][rng.random((V, V)) < 0.5] = 0.0
    comm = np.zeros((B, C, 3), dtype=np.int64)
    paths = np.full((B, C, PL), -1, dtype=np.int32)
    plen = np.zeros((B, C), dtype=np.int32)
    for b in range(B):
        for c in range(C):
            s, d = rng.choice(V, size=2, replace=False)
            comm[b, c] = (s, d, rng.integers(1, 500))
            n = int(rng.integers(2, min(V, PL) + 1))
            nodes = [s] + [x for x in rng.permutation(V) if x not in (s, d)][: n - 2]
            if rng.random() > 0.2:
                nodes.append(d)
            paths[b, c, :len(nodes)] = nodes
            plen[b, c] = len(nodes)
    return cap, comm, paths, plen


def _both(lib, L, dev, cap, comm, paths, plen):
    B, C, PL = paths.shape
    V = cap.shape[1]
    t = lambda a: torch.from_numpy(a).to(dev)
    capd, commd, pd, ld = t(cap), t(comm), t(paths), t(plen)
    out = []
    for static in (False, True):
        max_lf = torch.full((B,), -7.0, dtype=torch.float32, device=dev)
        complete = torch.full((B, C), 9, dtype=torch.uint8, device=dev)
        if static:
            rc = lib.gcn_load_factor_static(B, V, C, PL, L.ptr(pd), L.ptr(ld), L.ptr(capd), L.ptr(commd), L.ptr(max_lf),
                                            L.ptr(complete), L.stream_ptr(dev))
        else:
            usage = torch.empty((B, V, V), dtype=torch.int32, device=dev)
            rc = lib.gcn_load_factor(B, V, C, PL, L.ptr(pd), L.ptr(ld), L.ptr(capd), L.ptr(commd), L.ptr(usage),
                                     L.ptr(max_lf), L.ptr(complete), L.stream_ptr(dev))
        L.check(rc, "gcn_load_factor[_static]")
        torch.cuda.synchronize()
        out.append((max_lf.cpu().numpy(), complete.cpu().numpy()))
    return out


def test_static_scratch_matches_caller_scratch_twice():
    dev = _dev()
    from src.gcn.hip import lib as L
    lib = L.load()
    rng = np.random.default_rng(5)
    for B, V, C in ((6, 40, 7), (3, 14, 4)):          # the second call is smaller and sees the first call's counts
        PL = max(V, 10)
        cap, comm, paths, plen = _random_paths(rng, B, V, C, PL)
        (lf_a, done_a), (lf_b, done_b) = _both(lib, L, dev, cap, comm, paths, plen)
        assert np.isinf(lf_a).any() and np.isfinite(lf_a).any()
        assert lf_a.tobytes() == lf_b.tobytes()
        assert np.array_equal(done_a, done_b) and set(np.unique(done_a)) <= {0, 1}


def test_static_scratch_rejects_oversize():
    dev = _dev()
    from src.gcn.hip import lib as L
    lib = L.load()
    z = torch.zeros(16, dtype=torch.int32, device=dev)
    rc = lib.gcn_load_factor_static(2, 1024, 1, 1024, L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z), L.ptr(z),
                                    L.stream_ptr(dev))
    assert rc == -2 and b"exceeds the static scratch" in lib.gcn_last_error()
