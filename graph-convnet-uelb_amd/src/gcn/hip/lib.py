"""ctypes binding of libgcn_uelb_hip.so (the C ABI declared in include/gcn_uelb_hip.h).

PyTorch is used for device memory and streams only; every computation of the hot path runs in the
hand-written gfx950 kernels behind these entry points.  There is no fallback: if the library is
missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_void_p

import torch

_PKG_ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..", ".."))
LIB_PATH = os.environ.get("GCN_UELB_LIB") or os.path.join(_PKG_ROOT, "lib", "libgcn_uelb_hip.so")
ABI_VERSION = 3
MAX_H = 320
LAYER_NPARAMS = 12


class GcnHipError(RuntimeError):
    pass


BUILD_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def source_files():
    """The files the library is built from: csrc/*.hip, csrc/*.hpp and the public header."""
    csrc = os.path.join(_PKG_ROOT, "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".hpp")))
    files.append(os.path.abspath(os.path.join(_PKG_ROOT, "..", "include", "gcn_uelb_hip.h")))
    return files


def source_hash():
    """SHA-256 over the build flags and the names and contents of the library's sources.  build() stores it next
    to the library (``<lib>.srchash``); load() refuses a library whose stored hash differs from the sources in the
    tree, so new sources can never run against an old binary unnoticed (file times play no part)."""
    import hashlib
    h = hashlib.sha256(" ".join(BUILD_FLAGS).encode())
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


class ModelDesc(ctypes.Structure):
    """Mirror of ``gcn_model_desc`` (include/gcn_uelb_hip.h)."""
    _fields_ = [
        ("B", c_int), ("V", c_int), ("C", c_int), ("H", c_int), ("L", c_int),
        ("mlp_layers", c_int), ("voc_out", c_int), ("n_emb", c_int), ("aggregation", c_int),
        ("training", c_int), ("mask_invalid_edges", c_int), ("has_labels", c_int),
        ("p_layer", c_float), ("p_mlp", c_float), ("advance_rng", c_int),
    ]


_lib = None


def _declare(lib):
    lib.gcn_abi_version.restype = c_int
    lib.gcn_last_error.restype = c_char_p
    lib.gcn_prof_begin.restype = c_int
    lib.gcn_prof_begin.argtypes = [c_char_p]
    lib.gcn_prof_end.restype = c_int
    lib.gcn_prof_end.argtypes = [POINTER(ctypes.c_double), POINTER(c_int)]
    for name in ("gcn_layer_saved_floats", "gcn_layer_ws_bytes"):
        fn = getattr(lib, name)
        fn.restype = c_size_t
        fn.argtypes = [c_int, c_int, c_int, c_int]
    lib.gcn_layer_fwd.restype = c_int
    lib.gcn_layer_fwd.argtypes = [c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_size_t, c_void_p]
    lib.gcn_layer_bwd.restype = c_int
    lib.gcn_layer_bwd.argtypes = [c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_int,
                                  c_void_p, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_void_p,
                                  POINTER(c_void_p), c_void_p, c_size_t, c_void_p]
    lib.gcn_model_num_params.restype = c_int
    lib.gcn_model_num_params.argtypes = [POINTER(ModelDesc)]
    for name in ("gcn_model_saved_floats", "gcn_model_ws_bytes"):
        fn = getattr(lib, name)
        fn.restype = c_size_t
        fn.argtypes = [POINTER(ModelDesc)]
    lib.gcn_model_saved_layout.restype = c_int
    lib.gcn_model_saved_layout.argtypes = [POINTER(ModelDesc), c_int, POINTER(c_size_t)]
    lib.gcn_bn_relu_pattern.restype = c_int
    lib.gcn_bn_relu_pattern.argtypes = [c_size_t, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]
    lib.gcn_model_fwd.restype = c_int
    lib.gcn_model_fwd.argtypes = [POINTER(ModelDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_size_t, c_void_p]
    lib.gcn_model_bwd.restype = c_int
    lib.gcn_model_bwd.argtypes = [POINTER(ModelDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  POINTER(c_void_p), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  POINTER(c_void_p), c_void_p, c_size_t, c_void_p]
    lib.gcn_model_bwd_ev.restype = c_int
    lib.gcn_model_bwd_ev.argtypes = lib.gcn_model_bwd.argtypes + [POINTER(c_void_p)]
    lib.gcn_loss_ws_bytes.restype = c_size_t
    lib.gcn_loss_ws_bytes.argtypes = [c_int, c_int, c_int]
    lib.gcn_loss_edges_fwd.restype = c_int
    lib.gcn_loss_edges_fwd.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]
    lib.gcn_loss_edges_bwd.restype = c_int
    lib.gcn_loss_edges_bwd.argtypes = [c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p]
    if hasattr(lib, "gcn_beam_search"):
        lib.gcn_beam_search.restype = c_int
        lib.gcn_beam_search.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_size_t, c_void_p]
        lib.gcn_beam_ws_bytes.restype = c_size_t
        lib.gcn_beam_ws_bytes.argtypes = [c_int, c_int, c_int, c_int]
    if hasattr(lib, "gcn_beam_scores"):
        lib.gcn_beam_scores.restype = c_int
        lib.gcn_beam_scores.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    if hasattr(lib, "gcn_load_factor"):
        lib.gcn_load_factor.restype = c_int
        lib.gcn_load_factor.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]
    lib.gcn_load_factor_static.restype = c_int
    lib.gcn_load_factor_static.argtypes = [c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p]
    lib.gcn_layer_plan.restype = c_int
    lib.gcn_layer_plan.argtypes = [c_int, c_int, c_int, c_int, POINTER(c_int)]
    lib.gcn_path_sample.restype = c_int
    lib.gcn_path_sample.argtypes = [c_int, c_int, c_int, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.gcn_path_sample_bwd.restype = c_int
    lib.gcn_path_sample_bwd.argtypes = [c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]
    lib.gcn_reachability.restype = c_int
    lib.gcn_reachability.argtypes = [c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.gcn_entropy_grid_fwd.restype = c_int
    lib.gcn_entropy_grid_fwd.argtypes = [c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.gcn_entropy_grid_bwd.restype = c_int
    lib.gcn_entropy_grid_bwd.argtypes = [c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_void_p]
    lib.gcn_unpack_batch.restype = c_int
    lib.gcn_unpack_batch.argtypes = [c_int, c_int, c_int] + [c_void_p] * 10
    if hasattr(lib, "gcn_adam_step"):
        lib.gcn_adam_step.restype = c_int
        lib.gcn_adam_step.argtypes = [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_longlong, c_void_p]


def load():
    """Load the shared library (once).  Raises GcnHipError if it is missing or of another ABI."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GcnHipError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py build` "
                              f"(hipcc --offload-arch=gfx950); there is no fallback path")
        stamp = LIB_PATH + ".srchash"
        if not os.path.exists(stamp) and os.path.isdir(os.path.join(_PKG_ROOT, "csrc")):
            import warnings
            warnings.warn(f"{LIB_PATH} carries no source stamp ({stamp} is missing): it cannot be told from a library "
                          f"built from older sources; `python __graft_entry__.py build` writes the stamp", RuntimeWarning)
        if os.path.exists(stamp) and os.path.isdir(os.path.join(_PKG_ROOT, "csrc")):
            with open(stamp) as fh:
                built_from = (fh.read().split() or [''])[0]      # first word: the hash; build flags may follow
            if built_from != source_hash():
                raise GcnHipError(f"{LIB_PATH} was built from other sources than those in {_PKG_ROOT}/csrc "
                                  f"(stored hash {built_from[:12]}, tree {source_hash()[:12]}): rebuild it with "
                                  f"`python __graft_entry__.py build`")
        lib = ctypes.CDLL(LIB_PATH)
        _declare(lib)
        if lib.gcn_abi_version() != ABI_VERSION:
            raise GcnHipError(f"ABI mismatch: library {lib.gcn_abi_version()} vs binding {ABI_VERSION}")
        _lib = lib
    return _lib


def layer_plan(B, V, C, H):
    """The library's launch plan for a problem size: dict with the row-chunk length of the edge backward pass
    (``row_chunk``), chunks per graph, j tiles, edge-kernel grids, whether the wide kernels run and whether a ragged last
    j tile gets the packed-tile launch of the wide backward pass."""
    out = (c_int * 8)()
    check(load().gcn_layer_plan(B, V, C, H, out), "gcn_layer_plan")
    return dict(row_chunk=out[0], n_chunks=out[1], j_tiles=out[2], grid_fwd=out[3], grid_bwd=out[4], wide=bool(out[5]),
                two_blocks=bool(out[6]), packed_last_tile=bool(out[7]))


def check(rc, what):
    if rc != 0:
        msg = load().gcn_last_error().decode("utf-8", "replace")
        raise GcnHipError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def ptr_array(tensors):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


def stream_ptr(device):
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(t, name):
    if not t.is_cuda:
        raise GcnHipError(f"{name} is on {t.device}: the gfx950 HIP path needs tensors on a ROCm GPU "
                          f"(there is no CPU fallback; use the oracle under oracle/ for CPU checks)")


def f32c(t, name):
    """Contiguous fp32 view/copy of a tensor on its device."""
    require_cuda(t, name)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# workspace and dropout-state management (plumbing)
# ------------------------------------------------------------------------------------------------
_ws = {}
_rng = {}


def workspace(device, nbytes):
    """A per-device scratch buffer, grown on demand.  Calls that share it must be stream-ordered
    (they are: one host thread per device, current stream)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws.get(key)
    if torch.cuda.is_current_stream_capturing():
        # memory used inside a captured graph must stay valid for every replay: give each capture its own
        return torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def base_seed():
    """The seed of every counter-based generator of the library (dropout masks, path sampling): torch's initial seed
    with the rank of this process mixed in, so that data-parallel ranks seeded alike still draw different masks and
    paths for their different graphs.  63 bits (stored in int64 tensors)."""
    rank = 0
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank = torch.distributed.get_rank()
    seed = torch.initial_seed() if _seed_override is None else _seed_override
    return (seed + rank * 0x9E3779B97F4A7C15) & 0x7FFFFFFFFFFFFFFF


_seed_override = None      # set by reseed(seed): replaces torch's initial seed until reseed() is called without one


_reseed_hooks = []
_step_states = []          # weak references to {seed, step} tensors owned by HipTrainStep objects


def register_reseed_hook(fn):
    _reseed_hooks.append(fn)


def register_step_state(t):
    import weakref
    _step_states.append(weakref.ref(t))


def _dev_index(device):
    if device is None:
        return None
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def rng_state(device):
    """Device-resident {seed, step} words read by the dropout code of the kernels."""
    key = _dev_index(device)
    st = _rng.get(key)
    if st is None:
        st = torch.tensor([base_seed(), 0], dtype=torch.int64, device=device)
        _rng[key] = st
    return st


def reseed(seed=None, device=None):
    """Restart every generator of the library from ``seed`` (default: what the last ``torch.manual_seed`` set;
    either way with the rank mixed in, see ``base_seed``): the dropout {seed, step} words of the autograd path and of every live HipTrainStep
    (on ``device``, or on all devices), and the call counters of the path sampler.  ``torch.manual_seed(s)`` followed
    by ``reseed()`` reproduces a run; ``torch.manual_seed`` alone does not rewind the step and call counters."""
    global _seed_override
    _seed_override = None if seed is None else int(seed)
    seed = base_seed()
    want = _dev_index(device)
    fresh = torch.tensor([seed, 0], dtype=torch.int64)
    for key, st in list(_rng.items()):
        if want is None or key == want:
            st.copy_(fresh)
    alive = []
    for ref in _step_states:
        t = ref()
        if t is not None:
            alive.append(ref)
            if want is None or _dev_index(t.device) == want:
                t.copy_(fresh)
    _step_states[:] = alive
    for fn in _reseed_hooks:
        fn()


def next_rng_snapshot(device):
    """Advance the step word and return a private copy {seed, step} for one forward/backward pair."""
    st = rng_state(device)
    st[1:2].add_(1)
    return st.clone()
