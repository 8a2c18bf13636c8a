"""Host -> device batch path (reference: chassis.py:24-35 `GPULoaderIter`, which does a
synchronous `.to(device)` per item on the compute stream).

DevicePrefetcher keeps `depth` batches in flight: every tensor is staged in a reusable PINNED buffer and
copied with `non_blocking=True` on a dedicated copy stream while the previous step computes; the consumer's
stream waits on the copy's event only (no host sync).  With `jitter=DeviceJitter(...)` the jitter indices
(4th item) are generated on the device for the batch's (B, frames) instead of being shipped from the host
(the reference builds them element by element with numpy.random.choice in the collate function,
data.py:232-233).  With `mfcc=DeviceMfcc(...)` the conditioning input `mel` (2nd item) is computed on the device from
the staged wav windows (AEW_OP_MFCC) instead of by librosa in the collate function (data.py:230, mfcc.py:39-76); the
host then only ships the wav bytes and the batch's 2nd item may be None.

    for wav, mel, voice, jitter, *rest in DevicePrefetcher(loader, "cuda:0", jitter=DeviceJitter(0.12, seed)):
        pred, target, loss = model.run(wav, mel, voice, jitter)
"""
from __future__ import annotations

from collections import deque
from typing import Iterable, Optional

import numpy as np
import torch


class DevicePrefetcher:
    def __init__(self, batches: Iterable, device, depth: int = 2, jitter=None, mfcc=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("DevicePrefetcher stages batches for a cuda device")
        if depth < 1:
            raise ValueError("depth >= 1")
        self.it = iter(batches)
        self.depth = depth
        self.jitter = jitter
        self.mfcc = mfcc
        self.copy_stream = torch.cuda.Stream(self.device)
        self._pinned = [dict() for _ in range(depth + 1)]      # slot -> {(item index): pinned tensor}
        self._slot_ev = [None] * (depth + 1)                   # event after the slot's last H2D copies were issued
        self._slot = 0
        self._q = deque()

    def _stage(self, slot, i, t):
        buf = self._pinned[slot].get(i)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
            self._pinned[slot][i] = buf
        # plain host memcpy.  (torch's copy_ into a PINNED tensor goes through the HIP runtime here and blocks until the
        # device is idle: 6 ms per call measured with a training step in flight, which serialised host and GPU and made
        # the step through this loader 15 ms instead of 8)
        if t.device.type == "cpu" and t.is_contiguous() and t.dtype == buf.dtype and not t.requires_grad \
                and t.dtype != torch.bfloat16:                    # (numpy has no bfloat16)
            np.copyto(buf.numpy(), t.numpy())
        else:
            buf.copy_(t)
        return buf

    def _issue(self) -> bool:
        try:
            items = next(self.it)
        except StopIteration:
            return False
        slot = self._slot
        self._slot = (self._slot + 1) % len(self._pinned)
        if self._slot_ev[slot] is not None:
            # the staging buffers of this slot are about to be overwritten by a plain host memcpy: the H2D copies that
            # read them last time must have completed (they have, unless the copy stream lags depth + 1 batches behind)
            self._slot_ev[slot].synchronize()
        out = []
        with torch.cuda.stream(self.copy_stream):
            for i, x in enumerate(items):
                skip = (self.jitter is not None and i == 3) or (self.mfcc is not None and i == 1)
                if torch.is_tensor(x) and not skip:
                    out.append(self._stage(slot, i, x).to(self.device, non_blocking=True))
                else:
                    out.append(x)
            if self.mfcc is not None:
                out[1] = self.mfcc(out[0].float())                # on the copy stream, behind the wav copy
            if self.jitter is not None and len(items) > 3:
                mel = out[1]
                out[3] = self.jitter(mel.shape[0], mel.shape[2], self.device)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._slot_ev[slot] = ev
        self._q.append((tuple(out), ev))
        return True

    def __iter__(self):
        return self

    def __next__(self):
        while len(self._q) < self.depth and self._issue():
            pass
        if not self._q:
            raise StopIteration
        items, ev = self._q.popleft()
        torch.cuda.current_stream(self.device).wait_event(ev)
        for x in items:                                        # the consumer stream now owns these tensors
            if torch.is_tensor(x) and x.is_cuda:
                x.record_stream(torch.cuda.current_stream(self.device))
        self._issue()
        return items
