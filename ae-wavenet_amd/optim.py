"""Fused Adam over the flat parameter buffer (one kernel launch for all 23.6 M parameters).

Numerically torch.optim.Adam with its defaults (the reference's optimizer, checkpoint.py:49-50).
Drop-in for the harness: `ss.optim = FusedAdam(model, lr)` in place of `t.optim.Adam(...)`.
"""
from __future__ import annotations

import torch


class FusedAdam(torch.optim.Optimizer):
    """The moments live in the engine's flat buffers (adam.m / adam.v, same offsets as the
    parameters).  state_dict() / load_state_dict() speak torch.optim.Adam's format (per-parameter
    `step`, `exp_avg`, `exp_avg_sq`, parameters numbered in model.parameters() order) so that
    checkpoints interchange with the reference (checkpoint.py:61-63,87-98)."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.model = model
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self.grad_scale = grad_scale
        self._pending = None          # (step, {name: (exp_avg, exp_avg_sq)}) loaded before the engine exists

    @torch.no_grad()
    def step(self, closure=None):
        eng = self.model._engine
        if eng is None:
            raise RuntimeError("FusedAdam.step() before the first model.run()")
        self._flush_pending(eng)
        g = self.param_groups[0]
        eng.adam_step(g["lr"], self.grad_scale, g["betas"], g["eps"])

    # ---- torch.optim.Adam-compatible state -------------------------------------------------
    def _names(self):
        return [n for n, _ in self.model.named_parameters()]

    def _flush_pending(self, eng):
        if self._pending is None:
            return
        step, moments = self._pending
        for n, (m, v) in moments.items():
            o, k = eng.ps.off[n], eng.ps.numel_of(n)
            eng.adam_m[o:o + k].copy_(m.reshape(-1).to(eng.adam_m.device))
            eng.adam_v[o:o + k].copy_(v.reshape(-1).to(eng.adam_v.device))
        eng.step_count = int(step)
        self._pending = None

    def state_dict(self):
        names = self._names()
        eng = self.model._engine
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(names)))}
        state = {}
        if eng is not None and self._pending is None and eng.step_count > 0:
            torch.cuda.synchronize() if eng.adam_m.is_cuda else None
            for i, n in enumerate(names):
                o, k, shp = eng.ps.off[n], eng.ps.numel_of(n), eng.ps.shape[n]
                state[i] = {"step": torch.tensor(float(eng.step_count)),
                            "exp_avg": eng.adam_m[o:o + k].reshape(shp).detach().cpu().clone(),
                            "exp_avg_sq": eng.adam_v[o:o + k].reshape(shp).detach().cpu().clone()}
        elif self._pending is not None:
            step, moments = self._pending
            for i, n in enumerate(names):
                if n in moments:
                    state[i] = {"step": torch.tensor(float(step)), "exp_avg": moments[n][0].clone(),
                                "exp_avg_sq": moments[n][1].clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        names = self._names()
        groups = state_dict["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(names):
            raise ValueError(f"optimizer state has {len(order)} parameters, the model {len(names)}")
        g0 = groups[0]
        for k in ("lr", "betas", "eps"):
            if k in g0:
                self.param_groups[0][k] = tuple(g0[k]) if k == "betas" else g0[k]
        st = state_dict.get("state", {})
        moments, step = {}, 0
        for pos, idx in enumerate(order):
            s = st.get(idx, st.get(str(idx)))
            if s is None:
                continue
            p_shape = tuple(dict(self.model.named_parameters())[names[pos]].shape)
            if tuple(s["exp_avg"].shape) != p_shape:
                raise ValueError(f"exp_avg of parameter {names[pos]} has shape {tuple(s['exp_avg'].shape)}, expected {p_shape}")
            moments[names[pos]] = (s["exp_avg"].detach().float().cpu(), s["exp_avg_sq"].detach().float().cpu())
            step = max(step, int(float(s["step"])))
        self._pending = (step, moments) if moments else None
        eng = self.model._engine
        if eng is not None and self._pending is not None:
            self._flush_pending(eng)

    def zero_grad(self, set_to_none=True):
        # gradients are rewritten (not accumulated) by every backward; nothing to do
        return None
