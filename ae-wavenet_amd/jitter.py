"""Time-jitter indices generated on the device (reference: jitter.py:13-33, used by the collate
function data.py:232-233 on the CPU, one numpy.random.choice call per element).

    jit = DeviceJitter(replace_prob=0.12, seed=2507)
    jitter = jit(B, n_frames, device)         # int64 [B, n_frames], same meaning as the reference's Jitter(n)

The values follow the reference's distribution (at HEAD: iid offsets -1/0/+1 with probabilities
[p, 1-2p, p], first two entries the identity; `intended=True` selects the no-three-in-a-row rule its
docstring describes).  The random stream is the library's counter RNG, not numpy's MT19937: reproducible
from (seed, call number), independent of batch order and of the number of ranks when `seed` includes the
rank.  No CPU path: the HIP library is required.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class DeviceJitter:
    def __init__(self, replace_prob: float, seed: int = 0, intended: bool = False):
        if not 0.0 <= replace_prob <= 0.5:
            raise ValueError("replace_prob must be in [0, 0.5]")
        self.p, self.seed, self.mode = float(replace_prob), int(seed), int(bool(intended))
        self.calls = 0

    def op(self, out: torch.Tensor, B: int, n: int) -> L.Op:
        j = L.Jitter()
        j.out, j.out_pitch, j.B, j.n = out.data_ptr(), out.stride(0), B, n
        j.p, j.mode, j.seed, j.step = self.p, self.mode, self.seed & ((1 << 64) - 1), self.calls
        op = L.Op()
        op.kind = L.OP_JITTER
        op.u.jit = j
        return op

    def __call__(self, B: int, n: int, device) -> torch.Tensor:
        device = torch.device(device)
        if device.type != "cuda":
            raise L.AewError("DeviceJitter runs only through the HIP library on an MI355X")
        out = torch.empty(B, n, dtype=torch.int64, device=device)
        ops = (L.Op * 1)(self.op(out, B, n))
        fail = C.c_int(-1)
        with torch.cuda.device(device):
            st = torch.cuda.current_stream().cuda_stream
            L.check(L.load().aew_run_plan(C.cast(ops, C.c_void_p), 1, C.c_void_p(st), C.byref(fail)), "jitter")
        self.calls += 1
        return out
