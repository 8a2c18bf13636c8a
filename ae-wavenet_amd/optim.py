"""Fused Adam over the flat parameter buffer (one kernel launch for all 23.6 M parameters).

Numerically torch.optim.Adam with its defaults (the reference's optimizer, checkpoint.py:49-50).
Drop-in for the harness: `ss.optim = FusedAdam(model, lr)` in place of `t.optim.Adam(...)`.
"""
from __future__ import annotations

import torch


class FusedAdam(torch.optim.Optimizer):
    """The moments live in the engine's flat buffers (adam.m / adam.v, same offsets as the
    parameters); when no engine holds them (before the first run(), after model.to() / override() / a
    change of batch size) the model carries them (`HipModelBase._opt_carry`), so they survive every
    engine rebuild.  state_dict() / load_state_dict() speak torch.optim.Adam's format (per-parameter
    `step`, `exp_avg`, `exp_avg_sq`, parameters numbered in model.parameters() order) so that
    checkpoints interchange with the reference (checkpoint.py:61-63,87-98)."""

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
        dp = getattr(self.model, "_dp", None)
        if dp is not None and dp.sharded and not dp._solo():
            # data parallel, sharded: Adam on this rank's shards of the reduce-scattered gradient, then all-gather
            dp.optimizer_step(eng, g["lr"], self.grad_scale, betas=g["betas"], eps=g["eps"])
        else:
            eng.adam_step(g["lr"], self.grad_scale, g["betas"], g["eps"])

    # ---- torch.optim.Adam-compatible state -------------------------------------------------
    def _layout(self):
        """[(name, flat offset, numel, shape)] and the flat length: the ParamStore rule (engine.py), computed from
        the parameter specs so that it is known without an engine."""
        out, o = [], 0
        for name, p in self.model.named_parameters():
            k = p.numel()
            out.append((name, o, k, tuple(p.shape)))
            o += (k + 3) // 4 * 4
        return out, o

    def state_dict(self):
        lay, _ = self._layout()
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(lay)))}
        state = {}
        st = self.model._opt_state_flat()
        if st is not None and int(st[0]) > 0:
            step, m, v = st
            m, v = m.detach().cpu(), v.detach().cpu()           # (synchronises with the stream that wrote them)
            for i, (n, o, k, shp) in enumerate(lay):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": m[o:o + k].reshape(shp).clone(),
                            "exp_avg_sq": v[o:o + k].reshape(shp).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, state_dict):
        lay, total = self._layout()
        groups = state_dict["param_groups"]
        order = [i for g in groups for i in g["params"]]
        if len(order) != len(lay):
            raise ValueError(f"optimizer state has {len(order)} parameters, the model {len(lay)}")
        g0 = groups[0]
        for k in ("lr", "betas", "eps"):
            if k in g0:
                self.param_groups[0][k] = tuple(g0[k]) if k == "betas" else g0[k]
        st = state_dict.get("state", {})
        m, v = torch.zeros(total), torch.zeros(total)
        step, found = 0, False
        for pos, idx in enumerate(order):
            s = st.get(idx, st.get(str(idx)))
            if s is None:
                continue
            n, o, k, shp = lay[pos]
            if tuple(s["exp_avg"].shape) != shp:
                raise ValueError(f"exp_avg of parameter {n} has shape {tuple(s['exp_avg'].shape)}, expected {shp}")
            m[o:o + k] = s["exp_avg"].detach().float().cpu().reshape(-1)
            v[o:o + k] = s["exp_avg_sq"].detach().float().cpu().reshape(-1)
            step = max(step, int(float(s["step"])))
            found = True
        if not found:
            return
        model = self.model
        model._opt_carry = (step, m.to(model._device), v.to(model._device))
        model._opt_carry_partial = False
        dp = getattr(model, "_dp", None)
        if dp is not None:
            dp.moments_step = step                               # every rank loads the complete state
        eng = model._engine
        if eng is not None:
            eng.adam_m[:total].copy_(model._opt_carry[1])
            eng.adam_v[:total].copy_(model._opt_carry[2])
            eng.step_count = step

    def zero_grad(self, set_to_none=True):
        # the backward plan rewrites the flat gradient buffer: "cleared" is a flag the next backward reads (no memset;
        # without it the next backward adds to the existing .grad like any torch module, surface._grads_carried)
        self.model._grads_cleared = True
        if set_to_none:
            # like torch: nothing that looks at .grad between zero_grad() and the next backward sees last step's values
            # (the backward re-attaches the views into the flat gradient buffer, surface._after_backward)
            for p in self.model.parameters():
                p.grad = None
        else:
            # torch's set_to_none=False contract: .grad reads as zeros right away (the views point into the flat gradient
            # buffer, which the next backward rewrites anyway: one 95 MB fill, ~15 us)
            eng = getattr(self.model, "_engine", None)
            if eng is not None:
                eng.ps.grads[:eng.ps.numel].zero_()
        return None
