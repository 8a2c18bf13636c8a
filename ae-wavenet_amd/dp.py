"""Data parallelism: one process per GPU, RCCL over xGMI via torch.distributed.

The reference only distributes on TPU (xmp.spawn + xm.optimizer_step, train.py:58-60,
chassis.py:168-169).  Here every rank holds a full replica and an independent shard of audio
windows (sampler rule data.py:100-106); per step the ranks exchange
  * the gradients (SUM; the optimizer applies 1/world for mean-type losses — SURVEY §8e), and
  * the VQ-EMA statistics z_sum / n_sum (SUM) before the EMA update, so that all replicas keep
    an identical codebook (north-star requirement; the reference lets replicas drift).

Two schedules for the gradient exchange (same result up to fp32 summation order):
  train_step          all-reduce of the decoder tail under the encoder backward, all-reduce of the head under the
                      decoder's Adam range (round 1);
  train_step_sharded  reduce-scatter + SHARDED Adam + all-gather (what xm.optimizer_step amounts to on a ring):
                      every rank receives and updates only its 1/world shard of each region (a ring reduce-scatter moves
                      half the bytes of an all-reduce before the optimizer can start, and the Adam pass shrinks by
                      world), then the updated shards are all-gathered; the decoder region's reduce-scatter runs under the
                      encoder backward, and its all-gather - issued behind the head's - under the NEXT step's encoder
                      forward: forward() waits for the head's parameters before the first forward plan and for the
                      decoder's only between the two (TrainEngine.pack_dec_late).  Optional
                      bf16 gradient transport (half the xGMI bytes of the reduce-scatter; fp32 master parameters and
                      moments unchanged).
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


def shard_indices(n: int, rank: int, world: int, epoch: int) -> List[int]:
    """Replica-sharded looping sampler order of the reference (data.py:100-106)."""
    g = torch.Generator()
    g.manual_seed(epoch * world + rank)
    vals = list(range(rank, n, world))
    perm = torch.randperm(len(vals), generator=g).tolist()
    return [vals[i] for i in perm]


class DataParallel:
    def __init__(self, bucket_mb: float = 32.0, group=None, force_collectives: bool = False):
        """force_collectives: issue every collective also in a process group of ONE rank (they are identities there).  A
        one-GPU box can then put RCCL under the product's exact calls - views, dtypes, stream ordering against graph
        replays - which the world == 1 short cuts below would otherwise skip (tests/test_dp_gpu.py)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.force_collectives = bool(force_collectives)
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.sharded, self.bf16_grads = False, False
        self._pending, self._st = [], None
        self._bufs = {}                     # persistent transport buffers by (kind, region start, elements, dtype)
        self.moments_step = -1              # engine step count for which every rank holds ALL Adam moments
        self.timing = False                 # True: bracket every wait with events (exposed_ms())
        self._tev = []

    def _solo(self) -> bool:
        """One rank and no request to run the collectives anyway: every exchange is skipped."""
        return self.world == 1 and not self.force_collectives

    # ---- measurement: how long the compute stream actually stalls on collectives ---------------------------------
    def _wait(self, work, what: str):
        """work.wait() - with `timing` on, bracketed by events on the current stream: the elapsed time between them is
        the part of the collective that was NOT hidden under compute (nothing else runs between the two records)."""
        if not self.timing or not torch.cuda.is_available():
            work.wait()
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        work.wait()
        e1.record()
        self._tev.append((what, e0, e1))

    def exposed_ms(self, reset: bool = True):
        """{wait point: total exposed ms since the last reset} (synchronises)."""
        out = {}
        if self._tev:
            torch.cuda.synchronize()
            for what, e0, e1 in self._tev:
                out[what] = out.get(what, 0.0) + e0.elapsed_time(e1)
        if reset:
            self._tev = []
        return out

    def _buf(self, kind: str, start: int, numel: int, dtype, device) -> torch.Tensor:
        key = (kind, start, numel, dtype)
        t = self._bufs.get(key)
        if t is None or t.device != device:
            t = torch.empty(numel, dtype=dtype, device=device)
            self._bufs[key] = t
        return t

    # ---- the chained launches' sticky timeout word (TrainEngine.chain_guard) --------------------------------------------
    def _guard_sync(self, eng):
        """A hand-off wait that gave up on ONE rank poisons that rank's gradients - and through the exchange everybody's.
        The guard word the Adam / EMA ops read is therefore MAX-reduced over the ranks in front of the first gradient
        collective of a step (4 bytes, asynchronous, completes before the gradients do: same communicator, issue order),
        so every rank skips the update of such a step, not only the one that saw the timeout."""
        if self._solo() or getattr(eng, "chain_guard", None) is None or not eng._chain_flags():
            return
        self._guard_work = dist.all_reduce(eng.chain_guard[:1], op=dist.ReduceOp.MAX, group=self.group, async_op=True)

    def _guard_wait(self):
        w, self._guard_work = getattr(self, "_guard_work", None), None
        if w is not None:
            self._wait(w, "chain.guard")

    def _warn_merged_packs(self, eng):
        """TrainEngine.merge_packs = None resolves at engine construction from the torch.distributed state; an engine built
        BEFORE init_process_group packs every weight layout at the head of the first forward plan, and the decoder's
        parameter all-gather can then not stay in flight under the encoder forward (_late() is False).  Results are the
        same; the overlap is lost - say so once."""
        if self.world > 1 and self.sharded and getattr(eng, "merge_packs", False) and getattr(eng, "merge_packs_auto", False) \
                and not getattr(self, "_warned_packs", False):
            import warnings
            self._warned_packs = True
            warnings.warn("ae_wavenet_amd.dp: the engine was built before torch.distributed was initialised, so it packs all weight "
                          "layouts in ONE launch at the head of the forward (TrainEngine.merge_packs); the decoder's parameter "
                          "all-gather is then not overlapped with the encoder forward.  Build the model after "
                          "init_process_group, or set TrainEngine.merge_packs = False.")

    def grad_scale(self, mean_loss: bool) -> float:
        """Factor the optimizer applies to the summed gradient: 1/world reproduces the
        single-process gradient of a mean-type loss over the global batch; sum-type losses
        (VQ) keep the sum.  The VAE mixes both (mean NLL + annealed, clamped SUM of KL, vae_bn.py:90-125):
        prepare_vae() makes its KL part come out right under the 1/world factor."""
        return 1.0 / self.world if mean_loss else 1.0

    def prepare_vae(self, eng):
        """VAE under data parallel = the single-process objective over the global batch:
        mean(NLL over all ranks' positions) + anneal * clamp(sum of KL over ALL ranks' windows, min=free_nats).
        (1) the KL gradient coefficient is pre-multiplied by world (the optimizer divides the summed gradient by
        world for the mean-type NLL); (2) the free-nats gate of the clamp must see the GLOBAL KL: allreduce_kl()
        between forward and backward."""
        if eng.bn_type == "vae":
            eng.dp_world = self.world
            eng.set_anneal_weight(eng.anneal_weight)

    def allreduce_kl(self, eng):
        """Sum the KL value the backward's clamp gate reads (loss_buf[2]) over the ranks (VAE only)."""
        if eng.bn_type == "vae" and not self._solo():
            dist.all_reduce(eng.loss_buf[2:3], op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_grads(self, eng):
        flat = eng.ps.grads[:eng.ps.numel]
        if self._solo():
            return
        self._guard_sync(eng)
        self._guard_wait()
        n = flat.numel()
        for s in range(0, n, self.bucket_elems):
            dist.all_reduce(flat[s:s + self.bucket_elems], op=dist.ReduceOp.SUM, group=self.group)

    def backward_allreduce(self, eng):
        """Backward with the gradient exchange overlapped.  The flat gradient buffer is
        [encoder | bottleneck | decoder]; the decoder part (~40 % of the bytes at arch.vqvae-ema) is final
        once the decoder backward and its unpack are done, so its all-reduce is issued there
        (async, on the collective's own stream) and runs under the bottleneck / encoder backward.
        The head of the buffer follows when the backward is complete.  Numerically identical to
        backward() + allreduce_grads()."""
        if self._solo():
            eng.backward()
            return
        n, lo = eng.ps.numel, eng.dec_grad_offset
        flat = eng.ps.grads
        work = []

        def after_decoder():
            self._guard_sync(eng)
            work.append(dist.all_reduce(flat[lo:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        eng.backward(after_decoder=after_decoder)
        if lo > 0:
            work.append(dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._guard_wait()
        for w in work:
            w.wait()

    def train_step(self, eng, lr: float, grad_scale: float = 1.0, **adam_kw):
        """forward + backward + Adam with every collective off the critical path that can be:
          * EMA statistics: async all-reduce issued after the encoder / VQ part of the forward, consumed by the
            (deferred) EMA accumulation after the backward;
          * decoder gradients: async all-reduce between the two backward plans (under the encoder backward);
          * encoder gradients: all-reduce after the backward, under the Adam update of the decoder range."""
        if self._solo():
            eng.forward()
            eng.backward()
            eng.adam_step(lr, grad_scale, **adam_kw)
            return
        n, lo = eng.ps.numel, eng.dec_grad_offset
        flat = eng.ps.grads
        work = {}
        self.finish()                       # parameter all-gathers of a preceding sharded step

        def after_decoder():
            self._guard_sync(eng)
            work["dec"] = dist.all_reduce(flat[lo:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        eng.forward(self.allreduce_ema_async)
        self.allreduce_kl(eng)
        eng.backward(after_decoder=after_decoder)
        if lo > 0:
            work["enc"] = dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._guard_wait()
        self._wait(work["dec"], "grads.decoder")
        eng.adam_step(lr, grad_scale, lo=lo, hi=n, **adam_kw)
        if lo > 0:
            self._wait(work["enc"], "grads.encoder")
            eng.adam_step(lr, grad_scale, lo=0, hi=lo, count=False, **adam_kw)
        self.moments_step = eng.step_count  # all-reduce schedule: every rank updates everything

    # ---- reduce-scatter + sharded Adam + all-gather ------------------------------------------------------------
    def _split(self, a: int, b: int):
        """Region [a, b) of the flat buffer -> (shard elements s, remainder start): world * s elements are reduce-
        scattered (rank r owns [a + r*s, a + (r+1)*s)), the < 4*world elements from the remainder start are all-reduced
        and updated by every rank.  s is a multiple of 4 (the Adam kernel works on float4)."""
        s = ((b - a) // (4 * self.world)) * 4
        return s, a + self.world * s

    def _reduce_region(self, flat: torch.Tensor, a: int, b: int, bf16: bool):
        """Issue the reduction of gradient region [a, b): returns a list of async work handles and a `finish`
        callable to run after they completed (bf16 transport: copy the reduced shard back as fp32)."""
        s, rem = self._split(a, b)
        work, fin = [], []
        if s > 0:
            main = flat[a:a + self.world * s]
            mine = flat[a + self.rank * s: a + (self.rank + 1) * s]
            # persistent transport buffers (one set per region: no allocation in the step); out of place is valid on
            # every backend
            if bf16:
                half = self._buf("rs.in", a, self.world * s, torch.bfloat16, flat.device)
                out = self._buf("rs.out", a, s, torch.bfloat16, flat.device)
                half.copy_(main)                                 # fp32 -> bf16 (round to nearest even)
                work.append(dist.reduce_scatter_tensor(out, half, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                out = self._buf("rs.out", a, s, flat.dtype, flat.device)
                work.append(dist.reduce_scatter_tensor(out, main, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            fin.append(lambda: mine.copy_(out))
        if rem < b:
            work.append(dist.all_reduce(flat[rem:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return work, fin

    def _update_region(self, eng, a: int, b: int, lr, grad_scale, count, adam_kw, defer=None):
        """Adam on this rank's shard of region [a, b) (+ the replicated remainder), then the all-gather of the updated
        parameter shards (async; returned).  defer: a list - the all-gather is appended to it as a callable instead of
        being issued (collectives complete in issue order: optimizer_step issues the head's first when the decoder's
        may stay in flight under the next encoder forward)."""
        s, rem = self._split(a, b)
        work = []
        if s > 0:
            lo = a + self.rank * s
            eng.adam_step(lr, grad_scale, lo=lo, hi=lo + s, count=count, **adam_kw)
            count = False
            main = eng.ps.params[a:a + self.world * s]
            src = self._buf("ag.in", a, s, main.dtype, main.device)     # (in place all-gather is not valid on every backend)
            src.copy_(eng.ps.params[lo:lo + s])

            def gather():
                return dist.all_gather_into_tensor(main, src, group=self.group, async_op=True)
            if defer is None:
                work.append(gather())
            else:
                defer.append(gather)
        if rem < b:
            eng.adam_step(lr, grad_scale, lo=rem, hi=b, count=count, **adam_kw)
        return work

    def finish(self, region=None):
        """Wait for the parameter all-gathers of the previous sharded step.  region = None: all of them (before anything
        reads the parameters); "head": only the encoder / bottleneck region - what the first forward plan reads; the
        decoder's shards may stay in flight under it until forward(before_decoder=self.finish) (TrainEngine.pack_dec_late)."""
        keep = []
        for tag, w in getattr(self, "_pending", []):
            if region is None or tag == region:
                self._wait(w, "params.all_gather" + ("" if tag == "head" else ".decoder"))
            else:
                keep.append((tag, w))
        self._pending = keep

    @staticmethod
    def _late(eng) -> bool:
        """May the decoder's parameter all-gather stay in flight under the first forward plan?  Only if that plan reads no
        decoder parameter: the engine packs the decoder's layouts at the head of fwd_b (pack_dec_late, no merged pack) and
        there IS a head region.  ONE predicate for optimizer_step (issue order) and forward (wait points).  Engine-level
        readers of the parameters between steps (eng.conditioning(), eng.encode(), tools) must call finish() first - the
        module surface does."""
        return eng.dec_grad_offset > 0 and bool(getattr(eng, "pack_dec_late", False)) and not getattr(eng, "merge_packs", False)

    def forward(self, eng):
        """The forward of a sharded step: the encoder part starts as soon as ITS parameters are complete, the decoder's
        all-gather is waited for between the two forward plans."""
        late = self._late(eng)
        self._warn_merged_packs(eng)
        self.finish("head" if late else None)
        return eng.forward(self.allreduce_ema_async, before_decoder=self.finish if late else None)

    def backward_exchange(self, eng, bf16_grads: bool = False):
        """Backward with the sharded gradient exchange issued as the regions become final: decoder tail (reduce-scatter
        under the bottleneck / encoder backward), then the head.  optimizer_step() completes it."""
        n, lo = eng.ps.numel, eng.dec_grad_offset
        hi = getattr(eng, "dec_hi_offset", None)       # engines with two grouped wgrad launches: [hi, n) is final first
        flat = eng.ps.grads
        st = {"hi": hi}

        def after_decoder_hi():
            self._guard_sync(eng)
            st["dec_hi"] = self._reduce_region(flat, hi, n, bf16_grads)

        def after_decoder():
            if "dec_hi" not in st:
                self._guard_sync(eng)
            st["dec"] = self._reduce_region(flat, lo, n if "dec_hi" not in st else hi, bf16_grads)

        self.allreduce_kl(eng)
        if hi is not None:
            eng.backward(after_decoder=after_decoder, after_decoder_hi=after_decoder_hi)
        else:
            eng.backward(after_decoder=after_decoder)
        if "dec" not in st:
            after_decoder()
        st["head"] = self._reduce_region(flat, 0, lo, bf16_grads) if lo > 0 else ([], [])
        self._st = st

    def optimizer_step(self, eng, lr: float, grad_scale: float = 1.0, **adam_kw):
        """Sharded Adam + all-gather of the updated parameters (left in flight: finish() before the next forward)."""
        n, lo = eng.ps.numel, eng.dec_grad_offset
        st, self._st = self._st, None
        pend, counted, dec_end = [], True, n
        # late: the next forward waits for the head's parameters first and for the decoder's only between its two plans
        # (forward()); collectives complete in issue order, so the head's all-gather goes out first and the decoder's
        # right behind it.  Otherwise the decoder's all-gather is issued at once and runs under the head's Adam.
        late = self._late(eng)
        dec_gathers = [] if late else None
        self._guard_wait()
        if "dec_hi" in st:                               # the upper layers' region: reduced under the rest of the chain
            for w in st["dec_hi"][0]:
                self._wait(w, "grads.decoder_hi")
            for f in st["dec_hi"][1]:
                f()
            pend += [("dec", w) for w in self._update_region(eng, st["hi"], n, lr, grad_scale, True, adam_kw, dec_gathers)]
            counted, dec_end = False, st["hi"]
        for w in st["dec"][0]:
            self._wait(w, "grads.decoder")
        for f in st["dec"][1]:
            f()
        pend += [("dec", w) for w in self._update_region(eng, lo, dec_end, lr, grad_scale, counted, adam_kw, dec_gathers)]
        for w in st["head"][0]:
            self._wait(w, "grads.encoder")
        for f in st["head"][1]:
            f()
        if lo > 0:
            pend += [("head", w) for w in self._update_region(eng, 0, lo, lr, grad_scale, False, adam_kw)]
        for g in dec_gathers or []:
            pend.append(("dec", g()))
        self._pending = pend

    def train_step_sharded(self, eng, lr: float, grad_scale: float = 1.0, bf16_grads: bool = False, **adam_kw):
        """forward + backward + sharded optimizer step (see the module docstring).  Every rank ends with the same
        parameters as train_step() gives (fp32 transport: up to summation order; bf16 transport: the summed gradient
        is rounded to bf16 once per hop).  The Adam MOMENTS of a rank are valid for its own shards only (ZeRO-1): use
        gather_moments() before reading them (checkpoints)."""
        if self._solo():
            eng.forward()
            eng.backward()
            eng.adam_step(lr, grad_scale, **adam_kw)
            return
        self.forward(eng)
        self.backward_exchange(eng, bf16_grads)
        self.optimizer_step(eng, lr, grad_scale, **adam_kw)

    @staticmethod
    def _regions(eng):
        """The regions of the flat buffer the sharded schedule exchanges separately (each has its own shard layout):
        [upper decoder layers + post network |] decoder | encoder + bottleneck."""
        n, lo, hi = eng.ps.numel, eng.dec_grad_offset, getattr(eng, "dec_hi_offset", None)
        return [(hi, n), (lo, hi), (0, lo)] if hi is not None else [(lo, n), (0, lo)]

    def gather_moments(self, eng):
        """All-gather the Adam moments (each rank holds valid moments for its own shards only under the sharded step).
        COLLECTIVE: every rank must call it."""
        if self._solo():
            return
        for a, b in self._regions(eng):
            s, rem = self._split(a, b)
            if s > 0:
                for buf in (eng.adam_m, eng.adam_v):
                    dist.all_gather_into_tensor(buf[a:a + self.world * s], buf[a + self.rank * s: a + (self.rank + 1) * s].clone(),
                                                group=self.group)
        self.moments_step = eng.step_count

    def moments_complete(self, eng) -> bool:
        """Does this rank hold the Adam moments of ALL parameters for the engine's current step?  (Always under the
        all-reduce schedule; under the sharded one only right after gather_moments().)"""
        return self._solo() or not self.sharded or eng.step_count == 0 or self.moments_step == eng.step_count

    def sync_optimizer_state(self, model):
        """Make model / optimizer state readable on every rank: waits for the parameter all-gathers in flight and
        all-gathers the sharded Adam moments.  COLLECTIVE - call it on ALL ranks before any of them calls
        optimizer.state_dict() / checkpoint.save() (checkpoint.save does it when the model is attached)."""
        self.finish()
        eng = getattr(model, "_engine", None)
        if eng is not None and self.sharded and not self._solo() and not self.moments_complete(eng):
            self.gather_moments(eng)

    def allreduce_ema_async(self, z_sum: torch.Tensor, n_sum: torch.Tensor):
        """Like allreduce_ema but returns the work handle (the engine then defers the EMA accumulation)."""
        if self._solo():
            return None
        both = self._ema_flat(z_sum, n_sum)
        if both is None:
            self.allreduce_ema(z_sum, n_sum)
            return None
        return dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def _ema_flat(z_sum, n_sum):
        if z_sum.is_contiguous() and n_sum.is_contiguous() and \
                n_sum.data_ptr() == z_sum.data_ptr() + z_sum.numel() * z_sum.element_size() and \
                z_sum.untyped_storage().data_ptr() == n_sum.untyped_storage().data_ptr():
            return torch.as_strided(z_sum, (z_sum.numel() + n_sum.numel(),), (1,))
        return None

    def allreduce_ema(self, z_sum: torch.Tensor, n_sum: torch.Tensor):
        if self._solo():
            return
        # the engine allocates n_sum right behind z_sum: one collective instead of two
        both = self._ema_flat(z_sum, n_sum)
        if both is not None:
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group)
            return
        dist.all_reduce(z_sum, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(n_sum, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_params(self, eng, src: int = 0):
        self.finish()
        dist.broadcast(eng.ps.params, src, group=self.group)
        if eng.bn_type == "vqvae-ema":
            for t in (eng.emb, eng.ema_numer, eng.ema_denom):
                dist.broadcast(t, src, group=self.group)

    def attach(self, model, sharded: bool = False, bf16_grads: bool = False):
        """Hook the collectives into the module surface.  sharded=False: loss.backward() leaves the fully reduced
        gradients in every .grad (any optimizer).  sharded=True: backward issues the reduce-scatter, FusedAdam.step()
        updates this rank's shards and all-gathers the parameters (model.run waits for them); .grad then holds the
        reduced values for this rank's shards only, so only FusedAdam may be used."""
        model._dp = self
        self.sharded, self.bf16_grads = bool(sharded), bool(bf16_grads)
        model._ema_allreduce = self.allreduce_ema_async if sharded else self.allreduce_ema
        if getattr(model, "_engine", None) is not None:
            self.prepare_vae(model._engine)
