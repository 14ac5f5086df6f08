"""ctypes loader for oracle/beam_oracle.c (TEST INFRASTRUCTURE, never the product path)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
ALGO_ID = {"standard": 0, "deterministic": 1, "greedy": 2, "unconstrained": 3}


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libbeam_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        _LIB = ctypes.CDLL(path)
        _LIB.beam_oracle_search.restype = ctypes.c_int
    return _LIB


def search(y, cap, commodities, beam_size, algorithm="standard", mode_strict=False, max_iter=5, perms=None):
    """Same contract as oracle.beam_oracle.search but perms is an [n, C] int32 array.
    Returns (paths[B,C,PL] int32 padded with -1, path_len[B,C], feasible[B], attempts[B], perms_used)."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    cap = np.ascontiguousarray(cap, dtype=np.float32)
    comm = np.ascontiguousarray(commodities, dtype=np.int64)
    B, V, _, C = y.shape
    perms = np.zeros((0, C), np.int32) if perms is None else np.ascontiguousarray(perms, dtype=np.int32)
    PL = max(V, 10)
    paths = np.empty((B, C, PL), np.int32)
    plen = np.zeros((B, C), np.int32)
    feas = np.zeros(B, np.int32)
    att = np.zeros(B, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    used = _lib().beam_oracle_search(p(y), p(cap), p(comm), B, V, C, int(beam_size), ALGO_ID[algorithm],
                                     int(bool(mode_strict)), int(max_iter), p(perms), perms.shape[0],
                                     p(paths), p(plen), p(feas), p(att))
    if used < 0:
        raise RuntimeError("beam oracle ran out of permutations")
    return paths, plen, feas, att, used
