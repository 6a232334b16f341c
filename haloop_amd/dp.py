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


class ShardedUpdate:
    """The collectives of the sharded-optimizer data-parallel step ("prefer direct reduce-scatter + all-gather
    across the 7 links over ring"; the gradient semantics are DistributedDataParallel's, ha/attention_loop.py:154,203):

        reduce_scatter()   flat_grads[span of this rank] <- mean over ranks   (every rank receives 1/world of the reduced buffer)
        all_reduce_sum(t)  a few floats summed over ranks (the squared-norm partials of the clipped range: the clip needs the
                           GLOBAL gradient norm although each rank holds 1/world of the gradient)
        all_gather()       flat_params <- every rank's updated span

    between them each rank runs clip + AdamW on its span only (1/world of the optimizer's 28 bytes per parameter).  The flat
    buffers must be ``padded_numel(n, world)`` long so that the spans are equal and 16-byte aligned.  ``nccl`` (= RCCL) runs
    reduce_scatter_tensor / all_gather_into_tensor in place on the flat buffers; ``gloo`` (the CPU tests) has no reduce-scatter:
    there the mean is an all-reduce of which each rank keeps its span -- the same values."""

    def __init__(self, flat_params, flat_grads, group=None, always=False, wire_dtype='f32'):
        """always: issue the collectives on a one-rank group too (where each is the identity) -- how a single GPU rehearses the
        sharded step on the real backend, captured graphs included.
        wire_dtype: 'f32' reduce-scatters the fp32 gradients in place; 'bf16' rounds them to bf16 for the reduce-scatter (half of
        the step's gradient bytes on the links; the sums are formed in bf16 by the collective) and writes the fp32 mean of this
        rank's span back -- the all-gather of the parameters stays fp32 either way."""
        if wire_dtype not in ('f32', 'bf16'):
            raise ValueError(f"wire_dtype must be 'f32' or 'bf16', got {wire_dtype!r}")
        self.params, self.grads, self.group = flat_params, flat_grads, group
        self.always = bool(always) and dist.is_available() and dist.is_initialized()
        self.world = world_size(group)
        self.rank = dist.get_rank(group) if (self.world > 1 or self.always) else 0
        n = flat_grads.numel()
        if n % (4 * self.world) or flat_params.numel() != n:
            raise ValueError(f'flat buffers must hold padded_numel(n, world) elements, got {n} for world {self.world}')
        self.shard = n // self.world
        self.span = (self.rank * self.shard, (self.rank + 1) * self.shard)
        self._native = (self.world > 1 or self.always) and dist.get_backend(group) == 'nccl'
        self._avg_op = self._native and _avg_supported(flat_grads, group)
        self._wire = (torch.empty(n, dtype=torch.bfloat16, device=flat_grads.device)
                      if wire_dtype == 'bf16' and (self.world > 1 or self.always) else None)

    @staticmethod
    def padded_numel(n, world):
        q = 4 * world
        return (n + q - 1) // q * q

    def reduce_scatter(self):
        if self.world == 1 and not self.always:
            return
        lo, hi = self.span
        if self._wire is not None:
            w = self._wire
            if self.grads.is_cuda:
                from . import ops
                ops.cast_f32_to_bf16_(w, self.grads)
            else:
                w.copy_(self.grads)
            if self._native:
                dist.reduce_scatter_tensor(w[lo:hi], w, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
            if self.grads.is_cuda:
                ops.cast_bf16_to_f32_(self.grads[lo:hi], w[lo:hi], 1.0 / self.world)
            else:
                self.grads[lo:hi].copy_(w[lo:hi].float().mul_(1.0 / self.world))
            return
        if self._native:
            op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
            dist.reduce_scatter_tensor(self.grads[lo:hi], self.grads, op=op, group=self.group)     # in place: out = in + rank * count
            if not self._avg_op:
                self.grads[lo:hi].mul_(1.0 / self.world)
        else:
            dist.all_reduce(self.grads, op=dist.ReduceOp.SUM, group=self.group)
            self.grads[lo:hi].mul_(1.0 / self.world)

    def all_reduce_sum(self, t):
        if self.world > 1 or self.always:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather(self):
        if self.world == 1 and not self.always:
            return
        lo, hi = self.span
        if self._native:
            dist.all_gather_into_tensor(self.params, self.params[lo:hi], group=self.group)         # in place
        else:
            # (gloo copies into the list's tensors: contiguous slices of the flat buffer are written where they belong)
            dist.all_gather([self.params[r * self.shard:(r + 1) * self.shard] for r in range(self.world)], self.params[lo:hi].clone(),
                            group=self.group)
