"""Data-parallel plumbing for the utterance-sharded training step.

Semantics are those of the reference's only multi-GPU path, DistributedDataParallel in
ha/attention_loop.py:67-82,154,203: one process per GPU, rank-0 parameters broadcast once,
gradients SUMMED across ranks and divided by the world size once per optimizer step, skipped on
non-final accumulation micro-steps.  The flat fp32 gradient buffer of haloop_amd.train is reduced
in a few large buckets (xGMI is point-to-point: fewer, larger messages; RCCL picks the algorithm).
Backend-agnostic (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).
"""
import logging

import torch
import torch.distributed as dist

log = logging.getLogger(__name__)


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_parameters(flat_params, group=None, src=0):
    """DDP-constructor semantics: every rank starts from rank ``src``'s parameters."""
    if world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def shard_slice(global_batch, rank, world):
    """Utterances [rank*B_local, (rank+1)*B_local) of the global batch (equal shards)."""
    if global_batch % world:
        raise ValueError(f'global batch {global_batch} does not split evenly over {world} ranks')
    b = global_batch // world
    return slice(rank * b, (rank + 1) * b)


_AVG_PROBE = {}


def _avg_supported(like, group):
    """One tiny collective, identical on every rank, to learn whether ReduceOp.AVG works here."""
    key = id(group)
    if key not in _AVG_PROBE:
        try:
            t = torch.ones(4, device=like.device, dtype=like.dtype)
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=group)
            _AVG_PROBE[key] = bool(torch.allclose(t, torch.ones_like(t)))
        except Exception as e:                      # an old RCCL without AVG -- or a broken one: say so, SUM + scale still works
            log.warning('haloop_amd.dp: ReduceOp.AVG probe failed (%s: %s); averaging with SUM and a 1/world scale', type(e).__name__, e)
            _AVG_PROBE[key] = False
    return _AVG_PROBE[key]


class GradientAverager:
    """All-reduce(SUM)/world of ``flat_grads[span]`` in ``bucket_bytes`` pieces.

    ``start()`` launches the collectives asynchronously (they run on the backend's own stream, ordered
    after the work already enqueued on the current stream) and ``finish()`` waits for them and applies
    the 1/world scale, so a caller can keep enqueueing backward work in between (overlap).
    ``boundaries`` (element offsets) aligns bucket cuts with parameter groups.
    ``wire_dtype``: 'f32' reduces the fp32 buffer in place (DistributedDataParallel's semantics); 'bf16' rounds each bucket to
    bf16 for the collective (half the bytes on the links: ring all-reduce over xGMI is per-link bound) and writes the fp32
    average back -- an option, since 8 bf16 additions cost ~2 decimal digits of each gradient element."""

    def __init__(self, flat_grads, group=None, bucket_bytes=64 << 20, boundaries=None, span=None, wire_dtype='f32'):
        if wire_dtype not in ('f32', 'bf16'):
            raise ValueError(f"wire_dtype must be 'f32' or 'bf16', got {wire_dtype!r}")
        self.flat, self.group = flat_grads, group
        self.wire_dtype = wire_dtype
        self.world = world_size(group)
        lo, hi = span if span is not None else (0, flat_grads.numel())
        self.span = (lo, hi)
        per = max(1, bucket_bytes // flat_grads.element_size())
        cuts = sorted(set([lo, hi] + [b for b in (boundaries or []) if lo < b < hi]))
        self.buckets = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            for c in range(a, b, per):
                self.buckets.append((c, min(c + per, b)))
        # RCCL averages in the collective itself; gloo has no AVG (and an old RCCL might not): probe once
        self._avg_op = (self.world > 1 and wire_dtype == 'f32' and dist.get_backend(group) == 'nccl' and
                        _avg_supported(flat_grads, group))
        self._wire = None
        if self.world > 1 and wire_dtype == 'bf16':
            self._wire = [torch.empty(b - a, dtype=torch.bfloat16, device=flat_grads.device) for a, b in self.buckets]

    def _to_wire(self, i):
        a, b = self.buckets[i]
        if self.flat.is_cuda:
            from . import ops
            ops.cast_f32_to_bf16_(self._wire[i], self.flat[a:b])
        else:
            self._wire[i].copy_(self.flat[a:b])
        return self._wire[i]

    def _from_wire(self, i):
        a, b = self.buckets[i]
        if self.flat.is_cuda:
            from . import ops
            ops.cast_bf16_to_f32_(self.flat[a:b], self._wire[i], 1.0 / self.world)
        else:
            self.flat[a:b].copy_(self._wire[i].float().mul_(1.0 / self.world))

    def reduce_bucket(self, i, async_op=False):
        a, b = self.buckets[i]
        if self._wire is not None:
            return dist.all_reduce(self._to_wire(i), op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
        return dist.all_reduce(self.flat[a:b], op=op, group=self.group, async_op=async_op)

    def start(self):
        if self.world == 1:
            return []
        return [self.reduce_bucket(i, async_op=True) for i in range(len(self.buckets))]

    def finish(self, works):
        if self.world == 1:
            return
        for w in works:
            w.wait()
        if self._wire is not None:
            for i in range(len(self.buckets)):
                self._from_wire(i)
        elif not self._avg_op:
            lo, hi = self.span
            self.flat[lo:hi].mul_(1.0 / self.world)

    def average(self):
        self.finish(self.start())
