"""Drop-in for the transducer lattice of ha/transducer.py ([Graves12] Sequence Transduction with Recurrent Neural Networks) on the
HIP lattice kernels (csrc/lattice.hip).

``transducer_forward_score(joint, targets, joint_lengths, target_lengths)`` = ha/transducer.py:175-207: joint [N, T, U+1, K]
log-probabilities ((f + g).log_softmax(-1)), symbol 0 blank, targets [N, U] -> losses [N]; differentiable w.r.t. ``joint`` (alpha-beta
backward kernel where the reference uses autograd through its log-space scan).  ``transducer_forward_score4(joint, targets)``
(:145-172) is the single-sequence form.  Any T: the reference pads its scan to 2 ** round(log2(T)) (:194) and raises when that is
smaller than T; the recurrence it defines is computed here cell by cell along the anti-diagonals.  The probability-domain study
versions (transducer_forward_score1-3, :10-142) are not built.
"""
import torch

from . import _lib, ops


class _Transducer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, joint, targets, joint_lengths, target_lengths):
        j = joint.detach().float().contiguous()
        losses, workspace = ops.transducer_fwd(j, targets, joint_lengths, target_lengths, keep=joint.requires_grad)
        ctx.saved = (j, targets, joint_lengths, target_lengths, workspace, losses)
        return losses.clone()

    @staticmethod
    def backward(ctx, grad_losses):
        j, targets, joint_lengths, target_lengths, workspace, losses = ctx.saved
        return ops.transducer_bwd(j, targets, joint_lengths, target_lengths, workspace, losses, grad_losses.float().contiguous()), None, None, None


def transducer_forward_score(joint, targets, joint_lengths, target_lengths):
    """(N, T, U+1, K), (N, U), (N,), (N,) -> losses (N,)  (ha/transducer.py:175-207)."""
    if not joint.is_cuda:
        raise _lib.HaloError('haloop_amd.transducer.transducer_forward_score runs on the HIP device only (no CPU path)')
    dev = joint.device
    N, T, U1, K = joint.shape
    if targets.shape != (N, U1 - 1):
        raise ValueError(f'targets must be [N, U] = [{N}, {U1 - 1}], got {tuple(targets.shape)}')
    jl = joint_lengths.to(device=dev, dtype=torch.int32).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int32).contiguous()
    return _Transducer.apply(joint, targets.to(device=dev, dtype=torch.int64).contiguous(), jl, tl)


def transducer_forward_score4(joint, targets):
    """(T, U+1, K), (U,) -> scalar loss  (ha/transducer.py:145-172)."""
    T, U1, _ = joint.shape
    dev = joint.device
    return transducer_forward_score(joint[None], targets[None], torch.tensor([T], device=dev), torch.tensor([U1 - 1], device=dev))[0]
