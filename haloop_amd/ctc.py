"""Drop-in for ha/ctc.py: the three CTC forward-score functions and ctc_reduce_mean, each a single
launch of the wave-per-utterance alpha kernel with the flags that reproduce that variant."""
import torch

from . import _lib, ops

_SCORE3 = _lib.HALO_CTC_FULL_LATTICE | _lib.HALO_CTC_FINITE_MIN
_SCORE2 = _lib.HALO_CTC_FULL_LATTICE | _lib.HALO_CTC_NO_LEAD_BLANK_LOOP
_SCORE1 = _SCORE2 | _lib.HALO_CTC_WRAP_SKIP


def _single(emissions, targets, flags):
    T = emissions.shape[0]
    dev = emissions.device
    nll, _, _ = ops.ctc_fwd(emissions.float().contiguous()[:, None, :], True, targets[None].to(dev),
                            torch.tensor([T], device=dev), torch.tensor([targets.shape[0]], device=dev), flags)
    return nll[0]


def ctc_forward_score1(emissions, targets):
    """(T, C), (S,) -> scalar; ha/ctc.py:4-50 including its python-index wrap at s=1."""
    return _single(emissions, targets, _SCORE1)


def ctc_forward_score2(emissions, targets):
    """(T, C), (S,) -> scalar; ha/ctc.py:54-107."""
    return _single(emissions, targets, _SCORE2)


def ctc_forward_score3(emissions, targets, emission_lengths, target_lengths):
    """(T, N, C), (N, S), (N,), (N,) -> (N,); ha/ctc.py:110-174 (finfo.min as log zero)."""
    dev = emissions.device
    em = emissions.float()
    if em.stride(-1) != 1:
        em = em.contiguous()
    nll, _, _ = ops.ctc_fwd(em, True, targets.to(dev), emission_lengths.to(dev), target_lengths.to(dev), _SCORE3)
    return nll


def ctc_reduce_mean(losses, target_lengths):
    return (losses / target_lengths).mean(-1)
