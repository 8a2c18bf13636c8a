"""Data parallelism: one process per GPU, RCCL over xGMI via torch.distributed.

The reference only distributes on TPU (xmp.spawn + xm.optimizer_step, train.py:58-60,
chassis.py:168-169).  Here every rank holds a full replica and an independent shard of audio
windows (sampler rule data.py:100-106); per step the ranks exchange
  * the flat gradient buffer (one bucketed all-reduce, SUM; the optimizer applies 1/world for
    mean-type losses — SURVEY §8e), and
  * the VQ-EMA statistics z_sum / n_sum (SUM) before the EMA update, so that all replicas keep
    an identical codebook (north-star requirement; the reference lets replicas drift).
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
    def __init__(self, bucket_mb: float = 32.0, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)

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
        if eng.bn_type == "vae" and self.world > 1:
            dist.all_reduce(eng.loss_buf[2:3], op=dist.ReduceOp.SUM, group=self.group)

    def allreduce_grads(self, eng):
        flat = eng.ps.grads[:eng.ps.numel]
        if self.world == 1:
            return
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
        if self.world == 1:
            eng.backward()
            return
        n, lo = eng.ps.numel, eng.dec_grad_offset
        flat = eng.ps.grads
        work = []

        def after_decoder():
            work.append(dist.all_reduce(flat[lo:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

        eng.backward(after_decoder=after_decoder)
        if lo > 0:
            work.append(dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in work:
            w.wait()

    def train_step(self, eng, lr: float, grad_scale: float = 1.0, **adam_kw):
        """forward + backward + Adam with every collective off the critical path that can be:
          * EMA statistics: async all-reduce issued after the encoder / VQ part of the forward, consumed by the
            (deferred) EMA accumulation after the backward;
          * decoder gradients: async all-reduce between the two backward plans (under the encoder backward);
          * encoder gradients: all-reduce after the backward, under the Adam update of the decoder range."""
        if self.world == 1:
            eng.forward()
            eng.backward()
            eng.adam_step(lr, grad_scale, **adam_kw)
            return
        n, lo = eng.ps.numel, eng.dec_grad_offset
        flat = eng.ps.grads
        work = {}

        def after_decoder():
            work["dec"] = dist.all_reduce(flat[lo:n], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        eng.forward(self.allreduce_ema_async)
        self.allreduce_kl(eng)
        eng.backward(after_decoder=after_decoder)
        if lo > 0:
            work["enc"] = dist.all_reduce(flat[:lo], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        work["dec"].wait()
        eng.adam_step(lr, grad_scale, lo=lo, hi=n, **adam_kw)
        if lo > 0:
            work["enc"].wait()
            eng.adam_step(lr, grad_scale, lo=0, hi=lo, count=False, **adam_kw)

    def allreduce_ema_async(self, z_sum: torch.Tensor, n_sum: torch.Tensor):
        """Like allreduce_ema but returns the work handle (the engine then defers the EMA accumulation)."""
        if self.world == 1:
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
        if self.world == 1:
            return
        # the engine allocates n_sum right behind z_sum: one collective instead of two
        both = self._ema_flat(z_sum, n_sum)
        if both is not None:
            dist.all_reduce(both, op=dist.ReduceOp.SUM, group=self.group)
            return
        dist.all_reduce(z_sum, op=dist.ReduceOp.SUM, group=self.group)
        dist.all_reduce(n_sum, op=dist.ReduceOp.SUM, group=self.group)

    def broadcast_params(self, eng, src: int = 0):
        dist.broadcast(eng.ps.params, src, group=self.group)
        if eng.bn_type == "vqvae-ema":
            for t in (eng.emb, eng.ema_numer, eng.ema_denom):
                dist.broadcast(t, src, group=self.group)

    def attach(self, model):
        model._dp = self
        model._ema_allreduce = self.allreduce_ema
        if getattr(model, "_engine", None) is not None:
            self.prepare_vae(model._engine)
