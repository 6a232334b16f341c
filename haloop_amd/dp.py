"""Data-parallel plumbing for the utterance-sharded training step.

Semantics are those of the reference's only multi-GPU path, DistributedDataParallel in
ha/attention_loop.py:67-82,154,203: one process per GPU, rank-0 parameters broadcast once,
gradients SUMMED across ranks and divided by the world size once per optimizer step, skipped on
non-final accumulation micro-steps.  The flat fp32 gradient buffer of haloop_amd.train is reduced
in a few large buckets (xGMI is point-to-point: fewer, larger messages; RCCL picks the algorithm).
Backend-agnostic (``nccl`` = RCCL on the GPUs, ``gloo`` in the CPU tests).
"""
import torch
import torch.distributed as dist


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


class GradientAverager:
    """All-reduce(SUM)/world of a flat gradient buffer in ``bucket_bytes`` pieces.

    ``boundaries`` (element offsets) lets the caller align buckets with parameter groups so that
    a bucket can be launched as soon as its gradients are final (overlap with the rest of backward)."""

    def __init__(self, flat_grads, group=None, bucket_bytes=64 << 20, boundaries=None):
        self.flat, self.group = flat_grads, group
        self.world = world_size(group)
        n = flat_grads.numel()
        per = max(1, bucket_bytes // flat_grads.element_size())
        cuts = sorted(set([0, n] + [b for b in (boundaries or []) if 0 < b < n]))
        self.buckets = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            for a in range(lo, hi, per):
                self.buckets.append((a, min(a + per, hi)))

    def reduce_bucket(self, i, async_op=False):
        a, b = self.buckets[i]
        return dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def average(self):
        if self.world == 1:
            return
        works = [self.reduce_bucket(i, async_op=True) for i in range(len(self.buckets))]
        for w in works:
            w.wait()
        self.flat.mul_(1.0 / self.world)
