"""Drop-in for ha/star.py: star-CTC ([Pratap22] Star Temporal Classification) forward score on the HIP lattice kernels.

``star_ctc_forward_score(emissions, targets, emission_lengths, target_lengths, star_penalty=-0.5)`` returns the per-utterance
losses [N] of ha/star.py:65-166 and is differentiable w.r.t. ``emissions`` (the reference relies on autograd through its Python loop;
here the backward is an alpha-beta kernel, csrc/lattice.hip).  ``logsubexp`` / ``intersperse_stars`` / ``intersperse_blanks`` are the
reference's small tensor helpers (ha/star.py:4-62) as plain torch expressions -- data preparation, not on the hot path: the kernel
never builds the 2V-wide star emissions, it evaluates the two star values where a lattice state needs them.

Quirks kept: the emissions are widened with penalty 0 whatever ``star_penalty`` is (ha/star.py:82), the penalty applies to the
transitions INTO a star (:128), labels have no self loop (:129-130), a star can be re-entered from the blank after it (:117,128),
the recursion covers all T frames and the lengths only select the read-out (:153-162), "log 0" is finfo(float32).min (:91).
``animate=True`` (a debugging print loop, :139-141) is refused.
"""
import torch

from . import _lib, ops


def logsubexp(b, a):
    return b + torch.log1p(-torch.exp(a - b))


def intersperse_stars(log_probs, targets, penalty=0):
    """(T, N, V), (N, S) -> star log-probs (T, N, 2V) and star targets (N, 2S+1)  (ha/star.py:9-49)."""
    T, N, V = log_probs.shape
    complete_star_log_probs = log_probs[:, :, 1:].logsumexp(dim=-1, keepdim=True)
    star_log_probs = torch.cat([log_probs, complete_star_log_probs + penalty,
                                logsubexp(complete_star_log_probs, log_probs[:, :, 1:]) + penalty], dim=-1)
    star_targets = torch.stack([V + targets, targets], dim=1).mT.reshape(N, -1)
    star_targets = torch.cat([star_targets, targets.new(N, 1).fill_(V)], dim=-1)
    return star_log_probs, star_targets


def intersperse_blanks(targets, blank=0):
    """(N, S) -> (N, 2S+1): A B C -> _ A _ B _ C _  (ha/star.py:52-62)."""
    N, S = targets.shape
    out = torch.stack([torch.full_like(targets, blank), targets], dim=1).mT.reshape(N, -1)
    return torch.cat([out, targets.new_full((N, 1), blank)], dim=-1)


class _StarCtc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emissions, targets, emission_lengths, target_lengths, star_penalty):
        em = emissions.detach().float()
        if em.stride(-1) != 1:
            em = em.contiguous()
        losses, workspace = ops.star_ctc_fwd(em, targets, emission_lengths, target_lengths, star_penalty, keep=emissions.requires_grad)
        ctx.saved = (em, targets, emission_lengths, target_lengths, star_penalty, workspace, losses)
        return losses.clone()

    @staticmethod
    def backward(ctx, grad_losses):
        em, targets, emission_lengths, target_lengths, star_penalty, workspace, losses = ctx.saved
        grad = ops.star_ctc_bwd(em, targets, emission_lengths, target_lengths, star_penalty, workspace, losses,
                                grad_losses.float().contiguous())
        return grad, None, None, None, None


def star_ctc_forward_score(emissions, targets, emission_lengths, target_lengths, star_penalty=-0.5, animate=False):
    """(T, N, C) log-probabilities, (N, S), (N,), (N,) -> losses (N,)  (ha/star.py:65-166)."""
    if animate:
        raise NotImplementedError('animate=True is the reference\'s debugging print loop (ha/star.py:139-141): not built')
    if not emissions.is_cuda:
        raise _lib.HaloError('haloop_amd.star.star_ctc_forward_score runs on the HIP device only (no CPU path)')
    dev = emissions.device
    T = emissions.shape[0]
    el = emission_lengths.to(device=dev, dtype=torch.int64).contiguous()
    tl = target_lengths.to(device=dev, dtype=torch.int64).contiguous()
    return _StarCtc.apply(emissions, targets.to(device=dev, dtype=torch.int64).contiguous(), el, tl, float(star_penalty))
