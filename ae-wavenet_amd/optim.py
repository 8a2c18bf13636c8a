"""Fused Adam over the flat parameter buffer (one kernel launch for all 23.6 M parameters).

Numerically torch.optim.Adam with its defaults (the reference's optimizer, checkpoint.py:49-50).
Drop-in for the harness: `ss.optim = FusedAdam(model, lr)` in place of `t.optim.Adam(...)`.
"""
from __future__ import annotations

import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.model = model
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model._engine
        if eng is None:
            raise RuntimeError("FusedAdam.step() before the first model.run()")
        g = self.param_groups[0]
        eng.adam_step(g["lr"], self.grad_scale, g["betas"], g["eps"])

    def zero_grad(self, set_to_none=True):
        # gradients are rewritten (not accumulated) by every backward; nothing to do
        return None
