"""Checkpoint files in the reference's format.

The reference saves ONE torch.save()d dict (checkpoint.py:82-102):
    hps               hparams.Hyperparams (a dict subclass, pickled by class reference)
    epoch, step       dataset position
    optim_step        LR-schedule step (chassis.py:116-118)
    model_state_dict  nn.Module.state_dict() with the key names of SURVEY App. A.3
    optim             torch.optim.Adam.state_dict()
    rand_state, cuda_rand_states
and restores it in Checkpoint.__init__ (checkpoint.py:25-67): keys containing `_lead` /
`left_wing_size` are dropped, `load_state_dict(strict=False)`, `Adam.load_state_dict`.

`load()` reads such a file without the reference on sys.path (the pickled hps class is resolved to
config.Hyperparams); `restore()` applies it to a model of this package and its FusedAdam;
`save()` writes a file the reference's loader accepts (hps as a plain dict: the reference does
`Hyperparams(**ckpt['hps'])`, checkpoint.py:27-28).
"""
from __future__ import annotations

import pickle
from typing import Any, Dict, Optional

import torch

from . import config


class _Unpickler(pickle.Unpickler):
    """Resolves the reference's `hparams.Hyperparams` to this package's dict-with-attributes."""

    def find_class(self, module, name):
        if module == "hparams" and name == "Hyperparams":
            return config.Hyperparams
        return super().find_class(module, name)


class _PickleModule:
    """pickle-module facade for torch.load (it looks up Unpickler / load / loads on it)."""
    Unpickler = _Unpickler
    load = staticmethod(lambda f, **kw: _Unpickler(f, **kw).load())
    loads = staticmethod(pickle.loads)
    __name__ = "pickle"


def load(path, map_location="cpu") -> Dict[str, Any]:
    """Read a checkpoint written by the reference (or by save())."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)
    if not isinstance(ckpt, dict) or "model_state_dict" not in ckpt:
        raise ValueError(f"{path}: not an ae-wavenet checkpoint (no model_state_dict)")
    return ckpt


def filtered_state(ckpt: Dict[str, Any]) -> Dict[str, torch.Tensor]:
    """The key filter of checkpoint.py:54-55."""
    return {k: v for k, v in ckpt["model_state_dict"].items() if "_lead" not in k and "left_wing_size" not in k}


def restore(model, optim, ckpt: Dict[str, Any]) -> Dict[str, Any]:
    """Model weights (+ codebook buffers) and Adam state into `model` / `optim`
    (FusedAdam or torch.optim.Adam).  Returns the position fields the harness needs."""
    model.load_state_dict(filtered_state(ckpt), strict=False)
    if optim is not None and "optim" in ckpt:
        optim.load_state_dict(ckpt["optim"])
    out = {k: ckpt[k] for k in ("epoch", "step", "optim_step") if k in ckpt}
    if "hps" in ckpt:
        out["hps"] = config.Hyperparams(ckpt["hps"])
    return out


def save(path, model, optim, hps, epoch: int, step: int, optim_step: int,
         with_rng: bool = True) -> None:
    """Write the dict of checkpoint.py:87-98.  Tensors go to CPU, contiguous.

    Data parallel with the sharded optimizer (dp.attach(model, sharded=True)): every rank holds the Adam moments of its
    own shards only, so this call is COLLECTIVE - call it on ALL ranks; ranks that should not write pass path=None."""
    dp = getattr(model, "_dp", None)
    if dp is not None:
        dp.sync_optimizer_state(model)
    if path is None:
        return
    sd = {k: v.detach().to("cpu").contiguous().clone() for k, v in model.state_dict().items()}
    ostate = optim.state_dict()

    def cpu(o):
        if torch.is_tensor(o):
            return o.detach().to("cpu").contiguous().clone()
        if isinstance(o, dict):
            return {k: cpu(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(cpu(v) for v in o)
        return o

    state = {
        "hps": dict(hps),                      # plain dict: loads anywhere; the reference re-wraps it
        "epoch": int(epoch), "step": int(step), "optim_step": int(optim_step),
        "model_state_dict": sd, "optim": cpu(ostate),
        "rand_state": torch.get_rng_state() if with_rng else None,
        "cuda_rand_states": (torch.cuda.get_rng_state_all() if (with_rng and torch.cuda.is_available()) else None),
    }
    torch.save(state, path)
