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


class SpanSharded:
    """The sharded-optimizer data-parallel step with the exchange cut where the backward finishes its gradients and where the forward
    consumes its weights (DistributedDataParallel overlaps its buckets with backward, ha/attention_loop.py:154,203):

        flat buffers  =  [ small | big ],   big = [ late span | early span ]

    * ``early`` -- the top LSTM layer's two matrices, final when their weight-gradient launch retires: ``reduce_scatter('early')`` is
      started on a side stream at that point (``after`` = the event the library records there) and runs beside the lower layer's
      products and the front end's backward; ``late`` -- the lower layers' recurrent matrices, reduce-scattered behind the backward.
      Rank r owns chunk r of EACH span (fp32 master values, Adam moments, clip + AdamW on those chunks only).
    * ``small`` -- everything else (biases, front end, input projection, classifier: 5 % of the parameters): one all-reduce(AVG), every
      rank applies the identical update, nothing to gather.
    * ``all_gather()`` -- with ``gather_bf16`` (single-pass bf16 arithmetic, where every consumer of those matrices multiplies by their
      bf16 values) the owners send the bf16 roundings of their updated chunks, HALF the bytes of an fp32 all-gather, in ONE collective
      over a rank-major staging buffer, and every rank expands the others' records into its flat parameters (its own fp32 masters stay);
      ``gather_masters()`` exchanges the fp32 masters themselves (checkpoints, evaluation in another arithmetic).  Otherwise the
      chunks are all-gathered in fp32, in place.

    ``nccl`` (= RCCL) runs reduce_scatter_tensor / all_gather_into_tensor; under ``gloo`` (the CPU tests) a reduce-scatter is one
    ``reduce`` per owner -- only the owner receives the sum -- and the all-gather the list form."""

    def __init__(self, flat_params, flat_grads, early, late, small, group=None, always=False, gather_bf16=False):
        self.params, self.grads, self.group = flat_params, flat_grads, group
        self.always = bool(always) and dist.is_available() and dist.is_initialized()
        self.world = world_size(group)
        self.active = self.world > 1 or self.always
        self.rank = dist.get_rank(group) if self.active else 0
        self.small = tuple(small)
        self.spans = {}
        for name, (lo, hi) in (('late', late), ('early', early)):
            if hi > lo:
                if (hi - lo) % (4 * self.world) or lo % 4:
                    raise ValueError(f'span {name} [{lo}, {hi}) does not cut into {self.world} chunks of whole float4s')
                self.spans[name] = (lo, hi, (hi - lo) // self.world)
        self._native = self.active and dist.get_backend(group) == 'nccl'
        self._avg_op = self._native and _avg_supported(flat_grads, group)
        self.gather_bf16 = bool(gather_bf16) and bool(self.spans)
        self._per_rank = sum(c for _, _, c in self.spans.values())
        self._stage = (torch.empty(self.world * self._per_rank, dtype=torch.bfloat16, device=flat_params.device)
                       if self.gather_bf16 and self.active else None)
        self._side = torch.cuda.Stream(device=flat_grads.device) if flat_grads.is_cuda else None
        self.bytes_on_wire = None

    # ---- ownership ---------------------------------------------------------------------------
    def own(self, name):
        lo, _, c = self.spans[name]
        return (lo + self.rank * c, lo + (self.rank + 1) * c)

    def own_ranges(self):
        """What this rank updates: its chunk of every reduce-scattered span, and the replicated small range."""
        out = [self.own(n) for n in self.spans]
        if self.small[1] > self.small[0]:
            out.append(self.small)
        return sorted(out)

    def norm_ranges(self):
        """Ranges whose squared norms, summed over the ranks, give the whole buffer's: the replicated part counts on rank 0 only."""
        out = [self.own(n) for n in self.spans]
        if self.rank == 0 and self.small[1] > self.small[0]:
            out.append(self.small)
        return sorted(out)

    # ---- collectives -------------------------------------------------------------------------
    def reduce_scatter(self, name, after=None):
        """grads[own chunk of the span] <- mean over the ranks.  ``after``: a recorded torch.cuda.Event -- the collective is issued on a
        side stream that waits for it (instead of for everything enqueued so far); returns a handle for ``wait``."""
        if not self.active or name not in self.spans:
            return None
        lo, hi, c = self.spans[name]
        own_lo, own_hi = self.own(name)
        if self._native:
            op = dist.ReduceOp.AVG if self._avg_op else dist.ReduceOp.SUM
            if after is not None and self._side is not None:
                self._side.wait_event(after)
                with torch.cuda.stream(self._side):
                    work = dist.reduce_scatter_tensor(self.grads[own_lo:own_hi], self.grads[lo:hi], op=op, group=self.group, async_op=True)
                return ('side', work, name)
            dist.reduce_scatter_tensor(self.grads[own_lo:own_hi], self.grads[lo:hi], op=op, group=self.group)
            if not self._avg_op:
                self.grads[own_lo:own_hi].mul_(1.0 / self.world)
            return None
        # gloo: one reduce per owner; only the owner's buffer holds the sum afterwards
        for dst in range(self.world):
            piece = self.grads[lo + dst * c:lo + (dst + 1) * c]
            send = piece if dst == self.rank else piece.clone()
            dist.reduce(send, dst=dst, op=dist.ReduceOp.SUM, group=self.group)
        self.grads[own_lo:own_hi].mul_(1.0 / self.world)
        return None

    def wait(self, handle):
        """Order the current stream behind a side-stream collective started by ``reduce_scatter(after=...)``."""
        if handle is None:
            return
        _, work, name = handle
        work.wait()
        if not self._avg_op:
            own_lo, own_hi = self.own(name)
            self.grads[own_lo:own_hi].mul_(1.0 / self.world)

    def all_reduce_small(self):
        lo, hi = self.small
        if not self.active or hi <= lo:
            return
        if self._avg_op:
            dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(self.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            self.grads[lo:hi].mul_(1.0 / self.world)

    def all_reduce_sum(self, t):
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def _gather_span_f32(self, name):
        lo, hi, c = self.spans[name]
        own_lo, own_hi = self.own(name)
        if self._native:
            dist.all_gather_into_tensor(self.params[lo:hi], self.params[own_lo:own_hi], group=self.group)         # in place
        else:
            dist.all_gather([self.params[lo + r * c:lo + (r + 1) * c] for r in range(self.world)], self.params[own_lo:own_hi].clone(),
                            group=self.group)

    def all_gather(self):
        """Every rank's updated chunks to every rank: bf16 roundings through the staging buffer, or the fp32 values in place."""
        if not self.active or not self.spans:
            return
        if not self.gather_bf16:
            for name in self.spans:
                self._gather_span_f32(name)
            return
        names = list(self.spans)
        mine = self._stage[self.rank * self._per_rank:(self.rank + 1) * self._per_rank]
        if self.params.is_cuda:
            from . import ops
            ops.pack_ranges_bf16(self.params, [self.own(n) for n in names], mine)
        else:
            off = 0
            for n in names:
                a, b = self.own(n)
                mine[off:off + b - a].copy_(self.params[a:b])
                off += b - a
        if self._native:
            dist.all_gather_into_tensor(self._stage, mine, group=self.group)                                       # in place
        else:
            dist.all_gather([self._stage[r * self._per_rank:(r + 1) * self._per_rank] for r in range(self.world)], mine.clone(), group=self.group)
        if self.params.is_cuda:
            ops.expand_ranges_bf16(self._stage, [self.spans[n][0] for n in names], [self.spans[n][2] for n in names], self.world, self.rank,
                                   self.params)
        else:
            off = 0
            for n in names:
                lo, _, c = self.spans[n]
                for r in range(self.world):
                    if r != self.rank:
                        self.params[lo + r * c:lo + (r + 1) * c].copy_(self._stage[r * self._per_rank + off:r * self._per_rank + off + c])
                off += c

    def gather_masters(self):
        """fp32 all-gather of the sharded spans: after it every rank holds every owner's exact master values (checkpoints)."""
        if self.active:
            for name in self.spans:
                self._gather_span_f32(name)

    def wire_bytes(self):
        """Bytes this rank sends (= receives) per step, by collective: the model DESIGN.md section 7 prices."""
        w = self.world
        f = (w - 1) / w if w > 1 else 0.0
        out = {}
        for name, (lo, hi, c) in self.spans.items():
            out['reduce_scatter_' + name] = int(4 * (hi - lo) * f)
        out['all_reduce_small'] = int(2 * 4 * (self.small[1] - self.small[0]) * f)
        big = sum(hi - lo for lo, hi, _ in self.spans.values())
        out['all_gather'] = int((2 if self.gather_bf16 else 4) * big * f)
        return out


class _RawDeviceArray:
    """A window of device memory the library allocated (halo_dx_alloc) as a torch tensor: torch.as_tensor reads __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (nbytes,), 'typestr': '|u1', 'data': (ptr, False), 'version': 2}


class DirectExchange(SpanSharded):
    """``SpanSharded`` with every collective done by THIS library over HIP-IPC-mapped peer arenas instead of by RCCL (SURVEY.md section 5.8 /
    8e's "direct reduce-scatter + all-gather across the 7 links"; csrc/dp_direct.hip): a rank writes its 1/N pieces straight into the
    owners' arenas -- all peers at once, one point-to-point link each -- signals an epoch word per peer, and each owner waits (bounded)
    for its peers' words and reduces the received pieces in rank order.  Same spans, ownership, clip / AdamW ranges and bf16 staging of
    the all-gather as ``SpanSharded``; the bootstrap (one exchange of 64-byte IPC handles) goes over ``group`` -- any backend.

    One process per GPU on a node, or several processes on ONE GPU (the tests: correctness only).  The small replicated range is
    all-reduced as "everybody pushes to everybody, everybody sums in rank order" (one exchange; every rank sends the whole range to every
    peer: the cheaper form up to three ranks) or, from four ranks on, in two exchanges -- 1/N pieces to their owners, rank-order sums, the
    summed pieces to everybody -- which moves 2 (N-1)/N of the range per rank instead of N-1 times it.  Bit-identical on every rank either way."""

    KINDS = ('early', 'late', 'small', 'norm', 'gather', 'small2')

    def __init__(self, flat_params, flat_grads, early, late, small, group=None, gather_bf16=False, norm_parts=None):
        super().__init__(flat_params, flat_grads, early, late, small, group, always=False, gather_bf16=gather_bf16)
        if not flat_grads.is_cuda:
            raise ValueError('DirectExchange needs HIP device buffers (the CPU tests use SpanSharded over gloo)')
        from . import _lib
        import ctypes as C
        self._lib, self._C = _lib, C
        self._native = False                       # (no RCCL collectives to capture: the tail runs as eager launches, epochs are launch arguments)
        self.side_ok = True                        # the early reduce-scatter may start from the library's mid-backward event
        w = max(self.world, 1)
        self._norm_parts = int(norm_parts or _lib.HALO_SUMSQ_PARTS)
        lo_s, hi_s = self.small
        a16 = lambda n: (n + 15) // 16 * 16
        off = 4096                                 # [0, 4096): the flag blocks, 64 bytes per kind
        self._off = {}
        for name in ('early', 'late'):
            if name in self.spans:
                self._off[name] = off
                off += a16(w * self.spans[name][2] * 4)
        self._off['small'] = off
        n_small = hi_s - lo_s
        self._small_chunk = n_small // w if (w >= 4 and n_small > 0 and n_small % (4 * w) == 0) else 0      # > 0: the two-exchange form
        self._off['small2'] = off + a16(w * self._small_chunk * 4)       # (the summed pieces land behind the inbox, inside the same region)
        off += a16(w * max(n_small, 4) * 4)
        self._off['norm'] = off
        off += a16(w * self._norm_parts * 4)
        self._off['gather'] = off
        self._esz = 2 if self.gather_bf16 else 4
        off += a16(w * max(self._per_rank, 8) * self._esz)
        self._bytes = off
        handle = C.create_string_buffer(64)
        base = C.c_void_p()
        _lib.check(_lib.lib().halo_dx_alloc(self._bytes, C.byref(base), handle), 'halo_dx_alloc')
        self._base = base.value
        self._arena = torch.as_tensor(_RawDeviceArray(self._base, self._bytes), device=flat_grads.device)      # uint8 view of the own arena
        handles = [None] * w
        if self.active:
            dist.all_gather_object(handles, bytes(handle.raw), group=self.group)
        self._peer = [None] * w
        self._opened = []
        for r in range(w):
            if r == self.rank or not self.active:
                self._peer[r] = self._base
            else:
                p = C.c_void_p()
                _lib.check(_lib.lib().halo_dx_open(handles[r], C.byref(p)), 'halo_dx_open')
                self._peer[r] = p.value
                self._opened.append(p.value)
        self._peers = (C.c_void_p * w)(*self._peer)
        self._epoch = {k: 0 for k in self.KINDS}
        stage_bytes = w * self._per_rank * self._esz
        self._stage = self._arena[self._off['gather']:self._off['gather'] + stage_bytes].view(torch.bfloat16 if self.gather_bf16 else torch.float32) \
            if self._per_rank else None
        if self.active:
            dist.barrier(group=self.group)         # every arena is mapped everywhere before the first push

    def close(self):
        """Unmap the peers' arenas and free this rank's (idempotent; also run when the object is collected).  Call it on every rank behind
        the last step: a rank's last wait has seen every peer's last push into its arena, so nothing writes there any more."""
        if getattr(self, '_base', None):
            torch.cuda.synchronize()
        for p in getattr(self, '_opened', []):
            self._lib.lib().halo_dx_close(p)
        self._opened = []
        if getattr(self, '_base', None):
            self._arena = self._stage = None
            self._lib.lib().halo_dx_free(self._base)
            self._base = None

    def __del__(self):
        try:
            self.close()
        except Exception:              # (interpreter shutdown: the runtime may be gone already)
            pass

    # ---- the three launches of a collective -----------------------------------------------------
    def _stream(self):
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())

    def _exchange(self, kind, src_ptr, piece_bytes, stride_bytes, same):
        L, st = self._lib.lib(), self._stream()
        self._epoch[kind] += 1
        flag_off = 64 * self.KINDS.index(kind)
        self._lib.check(L.halo_dx_push(src_ptr, piece_bytes, stride_bytes, int(same), self._peers, self._off[kind], self.world, self.rank, st), 'halo_dx_push')
        self._lib.check(L.halo_dx_signal(self._peers, flag_off, self.world, self.rank, self._epoch[kind], st), 'halo_dx_signal')
        self._lib.check(L.halo_dx_wait(self._base + flag_off, self.world, self.rank, self._epoch[kind], st), 'halo_dx_wait')

    def _reduce(self, kind, own, scale):
        self._lib.check(self._lib.lib().halo_dx_reduce(own.data_ptr(), self._base + self._off[kind], own.numel(), self.world, self.rank, scale,
                                                       self._stream()), 'halo_dx_reduce')

    # ---- collectives (SpanSharded's interface) ----------------------------------------------------
    def reduce_scatter(self, name, after=None):
        if not self.active or name not in self.spans:
            return None
        lo, hi, c = self.spans[name]
        own_lo, own_hi = self.own(name)

        def run():
            self._exchange(name, self.grads[lo:hi].data_ptr(), c * 4, c * 4, False)
            self._reduce(name, self.grads[own_lo:own_hi], 1.0 / self.world)
        if after is not None and self._side is not None:
            self._side.wait_event(after)
            with torch.cuda.stream(self._side):
                run()
                done = torch.cuda.Event()
                done.record()
            return ('side', done, name)
        run()
        return None

    def wait(self, handle):
        if handle is not None:
            torch.cuda.current_stream().wait_event(handle[1])

    def all_reduce_small(self):
        lo, hi = self.small
        if not self.active or hi <= lo:
            return
        own = self.grads[lo:hi]
        c = self._small_chunk
        if not c:
            self._exchange('small', own.data_ptr(), (hi - lo) * 4, 0, True)
            self._reduce('small', own, 1.0 / self.world)
            return
        # piece p -> rank p, which sums the world's pieces in rank order; then every rank's summed piece -> everybody
        mine = own[self.rank * c:(self.rank + 1) * c]
        self._exchange('small', own.data_ptr(), c * 4, c * 4, False)
        self._reduce('small', mine, 1.0 / self.world)
        self._exchange('small2', mine.data_ptr(), c * 4, 0, True)
        got = self._arena[self._off['small2']:self._off['small2'] + self.world * c * 4].view(torch.float32)
        if self.rank > 0:
            own[:self.rank * c].copy_(got[:self.rank * c])
        if self.rank + 1 < self.world:
            own[(self.rank + 1) * c:].copy_(got[(self.rank + 1) * c:])

    def all_reduce_sum(self, t):
        if not self.active:
            return
        if t.numel() != self._norm_parts or t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f'DirectExchange.all_reduce_sum: {self._norm_parts} contiguous fp32 partials')
        self._exchange('norm', t.data_ptr(), t.numel() * 4, 0, True)
        self._reduce('norm', t, 1.0)

    def _gather(self, as_bf16):
        from . import ops
        names = list(self.spans)
        esz = 2 if as_bf16 else 4
        stage = self._stage if as_bf16 == self.gather_bf16 else self._stage32()
        mine = stage[self.rank * self._per_rank:(self.rank + 1) * self._per_rank]
        if as_bf16:
            ops.pack_ranges_bf16(self.params, [self.own(n) for n in names], mine)
        else:
            off = 0
            for n in names:
                a, b = self.own(n)
                mine[off:off + b - a].copy_(self.params[a:b])
                off += b - a
        # my record -> slot `rank` of every peer's staging buffer
        self._exchange('gather', mine.data_ptr(), self._per_rank * esz, 0, True)
        if as_bf16:
            ops.expand_ranges_bf16(stage, [self.spans[n][0] for n in names], [self.spans[n][2] for n in names], self.world, self.rank, self.params)
        else:
            off = 0
            for n in names:
                lo, _, c = self.spans[n]
                for r in range(self.world):
                    if r != self.rank:
                        self.params[lo + r * c:lo + (r + 1) * c].copy_(stage[r * self._per_rank + off:r * self._per_rank + off + c])
                off += c

    def _stage32(self):
        """fp32 staging for gather_masters() under a bf16 all-gather: a second arena region would need a second mapping on every peer, so
        the masters travel through RCCL / gloo instead."""
        raise NotImplementedError

    def all_gather(self):
        if self.active and self.spans:
            self._gather(self.gather_bf16)

    def gather_masters(self):
        if not self.active:
            return
        if not self.gather_bf16:
            self._gather(False)
            return
        for name in self.spans:                    # (a checkpoint-time exchange: the bootstrap group's own all-gather)
            lo, hi, c = self.spans[name]
            own_lo, own_hi = self.own(name)
            dist.all_gather([self.params[lo + r * c:lo + (r + 1) * c] for r in range(self.world)], self.params[own_lo:own_hi].clone(), group=self.group)
