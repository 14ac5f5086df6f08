"""Multi-tensor Adam on the HIP library: the whole parameter list is updated by one or two kernel launches
instead of torch.optim.Adam's per-tensor (or foreach) op chains.

Drop-in for ``torch.optim.Adam(params, lr=...)`` as the reference's trainer builds it
(``src/gcn/train/trainer.py:95``): same constructor arguments (no amsgrad / maximize / capturable), same
``param_groups`` (so ``update_learning_rate`` keeps working) and the same ``state_dict`` layout
(``step`` tensor, ``exp_avg``, ``exp_avg_sq``), so optimizer checkpoints move between the two.
"""
from __future__ import annotations

import ctypes

import torch

from . import lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr}")
        if eps < 0.0:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if weight_decay < 0.0:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for group in self.param_groups:
            ps, gs, ms, vs = [], [], [], []
            step_no = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                L.require_cuda(p, "parameter")
                if p.grad.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32:
                    raise TypeError("FusedAdam updates float32 parameters only")
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam needs contiguous parameters")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                n = int(st["step"].item())
                if step_no is None:
                    step_no = n
                elif n != step_no:
                    # parameters that joined the group at different times: update them on their own
                    self._launch(lib, group, [p], [p.grad.contiguous()], [st["exp_avg"]], [st["exp_avg_sq"]], n)
                    continue
                ps.append(p)
                gs.append(p.grad.contiguous())
                ms.append(st["exp_avg"])
                vs.append(st["exp_avg_sq"])
            if ps:
                self._launch(lib, group, ps, gs, ms, vs, step_no)
        return loss

    @staticmethod
    def _launch(lib, group, ps, gs, ms, vs, step_no):
        dev = ps[0].device
        with torch.cuda.device(dev):
            numel = (ctypes.c_longlong * len(ps))(*[p.numel() for p in ps])
            b1, b2 = group["betas"]
            rc = lib.gcn_adam_step(len(ps), L.ptr_array(ps), L.ptr_array(gs), L.ptr_array(ms), L.ptr_array(vs), numel,
                                   float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                   float(group["weight_decay"]), int(step_no), L.stream_ptr(dev))
            L.check(rc, "gcn_adam_step")
