"""Forward + backward of ``ResidualGatedGCNModel`` as two library calls on static buffers, optionally
replayed from a captured HIP graph.

This is the trainer's fast path: it bypasses the autograd engine (one ``gcn_model_fwd`` and one
``gcn_model_bwd`` per step), keeps every buffer alive between steps (no allocation in the loop), writes
all parameter gradients into ONE flat fp32 buffer (``flat_grad``; each ``p.grad`` is a view of it) so that
data-parallel training needs a single RCCL all-reduce per step, and reads the dropout step counter
from device memory so that a captured graph draws fresh masks on every replay.
"""
from __future__ import annotations

import ctypes

import torch

from . import lib as L
from .functions import AGG_ID


class HipTrainStep:
    def __init__(self, model, batch_size, device, use_graph=True, compute_loss=True):
        self.lib = L.load()
        self.model = model
        self.device = torch.device(device)
        self.use_graph = use_graph
        dev = self.device
        B, V, C = int(batch_size), model.num_nodes, model.num_commodities
        self.B, self.V, self.C = B, V, C
        self.params = [p for p in model.hip_parameters()]
        for p in self.params:
            L.require_cuda(p, "model parameter")
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise L.GcnHipError("model parameters must be contiguous fp32")
        # one flat gradient buffer; p.grad are views of it
        sizes = [p.numel() for p in self.params]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 3) // 4 * 4          # keep every view 16-byte aligned
        self.flat_grad = torch.zeros(tot, dtype=torch.float32, device=dev)
        self.grad_offsets, self.grad_total = offs, tot      # float offsets of the parameters inside flat_grad
        self.done_events = None                             # see set_done_events
        self.grad_views = [self.flat_grad[o:o + n].view_as(p) for o, n, p in zip(offs, sizes, self.params)]
        for p, g in zip(self.params, self.grad_views):
            p.grad = g
        # static inputs
        self.cap = torch.zeros((B, V, V), dtype=torch.float32, device=dev)
        self.demand = torch.zeros((B, C), dtype=torch.float32, device=dev)
        self.x_nodes = torch.zeros((B, V, C), dtype=torch.int64, device=dev)
        self.y_edges = torch.zeros((B, V, V, C), dtype=torch.int64, device=dev)
        self.edge_cw = torch.ones(2, dtype=torch.float32, device=dev)
        self.dloss = torch.ones(1, dtype=torch.float32, device=dev)
        self.compute_loss = compute_loss
        self._desc_training = None
        self._graph = None
        self._build_desc(model.training)
        d = self.desc
        self.saved = torch.empty(self.lib.gcn_model_saved_floats(ctypes.byref(d)), dtype=torch.float32, device=dev)
        self.ws = torch.empty(self.lib.gcn_model_ws_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
        self.y_pred = torch.empty((B, V, V, C, model.voc_edges_out), dtype=torch.float32, device=dev)
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        self.counts = torch.zeros(2 * B * C, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(B * C, dtype=torch.uint8, device=dev)
        self.rng = torch.tensor([L.base_seed(), 0], dtype=torch.int64, device=dev)
        L.register_step_state(self.rng)          # lib.reseed() restarts this step's dropout stream too
        self._param_ptrs = L.ptr_array(self.params)
        self._grad_ptrs = L.ptr_array(self.grad_views)

    def _build_desc(self, training):
        m = self.model
        layer_p = float(m.gcn_layers[0].dropout.p) if m.num_layers else 0.0
        self.desc = L.ModelDesc(B=self.B, V=self.V, C=self.C, H=m.hidden_dim, L=m.num_layers, mlp_layers=m.mlp_layers,
                                voc_out=m.voc_edges_out, n_emb=m.nodes_embedding.num_embeddings,
                                aggregation=AGG_ID[m.aggregation], training=int(bool(training)), mask_invalid_edges=0,
                                has_labels=int(bool(self.compute_loss)), p_layer=layer_p, p_mlp=float(m.mlp_edges.dropout_rate),
                                advance_rng=1)
        self._desc_training = bool(training)

    # ------------------------------------------------------------------ data
    def load_batch(self, cap, demand, x_nodes, y_edges, edge_cw):
        """Copy one batch into the static input buffers (tensors on any device, or numpy arrays)."""
        t = torch.as_tensor
        self.cap.copy_(t(cap), non_blocking=True)
        self.demand.copy_(t(demand), non_blocking=True)
        self.x_nodes.copy_(t(x_nodes), non_blocking=True)
        self.y_edges.copy_(t(y_edges), non_blocking=True)
        self.edge_cw.copy_(t(edge_cw).to(torch.float32), non_blocking=True)

    # ------------------------------------------------------------------ launches
    def _launch(self):
        d, dev, lib = self.desc, self.device, self.lib
        st = L.stream_ptr(dev)
        rc = lib.gcn_model_fwd(ctypes.byref(d), L.ptr(self.cap), L.ptr(self.demand), L.ptr(self.x_nodes),
                               L.ptr(self.y_edges), L.ptr(self.edge_cw), self._param_ptrs, L.ptr(self.rng),
                               L.ptr(self.y_pred), L.ptr(self.loss_out), L.ptr(self.counts), L.ptr(self.flags),
                               L.ptr(self.saved), L.ptr(self.ws), self.ws.numel(), st)
        L.check(rc, "gcn_model_fwd")
        rc = lib.gcn_model_bwd_ev(ctypes.byref(d), L.ptr(self.cap), L.ptr(self.demand), L.ptr(self.x_nodes),
                                  L.ptr(self.y_edges), L.ptr(self.edge_cw), self._param_ptrs, L.ptr(self.rng),
                                  L.ptr(self.saved), L.ptr(self.loss_out), L.ptr(self.dloss), None, self._grad_ptrs,
                                  L.ptr(self.ws), self.ws.numel(), st, self.done_events)
        L.check(rc, "gcn_model_bwd_ev")

    def layer_grad_ranges(self):
        """Float ranges [lo, hi) of flat_grad: entry k < L holds the gradients of conv layer k, entry L those of the edge
        MLP, entry L + 1 the embeddings (parameters are laid out embeddings, layers 0..L-1, MLP)."""
        nl = self.model.num_layers
        per = (len(self.params) - 3 - 2 * self.model.mlp_layers) // max(nl, 1)
        offs = self.grad_offsets + [self.grad_total]
        ranges = [(offs[3 + per * k], offs[3 + per * (k + 1)]) for k in range(nl)]
        ranges.append((offs[3 + per * nl], self.grad_total))
        ranges.append((0, offs[3]))
        return ranges

    def set_done_events(self, events):
        """events: None, or L + 1 entries (torch.cuda.Event or None): entry k is recorded by the backward pass once the
        gradients of conv layer k (k < L) or of the edge MLP (k = L) are final (gcn_model_bwd_ev).  Eager launches only:
        a captured graph is replayed as a whole."""
        if events is None:
            self.done_events = None
            return
        if self.use_graph:
            raise L.GcnHipError("completion events need eager launches (use_graph=False)")
        if len(events) != self.model.num_layers + 1:
            raise L.GcnHipError("one event per conv layer plus one for the edge MLP")
        arr = (ctypes.c_void_p * len(events))()
        with torch.cuda.device(self.device):
            for k, ev in enumerate(events):
                if ev is not None:
                    ev.record()                     # creates the underlying event
                    arr[k] = ev.cuda_event
        self._done_event_objs = list(events)        # keep them alive
        self.done_events = arr

    def _step_body(self):
        # new dropout masks every step: the forward call advances the step word of self.rng on the device
        # (advance_rng), so a replayed graph draws fresh masks without a launch of its own
        self._launch()

    def step(self):
        """One forward + backward on the batch currently in the static buffers.  Returns loss_out
        (device tensor [loss, sum w, edge error %, 0]); gradients are in flat_grad / p.grad."""
        if self._desc_training != bool(self.model.training):
            self._build_desc(self.model.training)
            self._graph = None
        with torch.cuda.device(self.device):
            if not self.use_graph:
                self._step_body()
            else:
                if self._graph is None:
                    self._capture()
                self._graph.replay()
        return self.loss_out

    def _capture(self):
        torch.cuda.synchronize(self.device)
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self._step_body()        # warm-up outside capture (lazy module loading, LDS attribute calls)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step_body()
        self._graph = g
