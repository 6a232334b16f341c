"""CPU restatement of haloop's sequence math: CTC forward scores, greedy collapse, beam search.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Small-case code: explicit Python loops
over time (and over beams), torch float32 scalars/tensors on the CPU so the arithmetic is the
same ATen arithmetic the reference runs.  Each function names the reference lines it follows
(paths relative to /root/reference).
"""
import math

import torch

BLANK = 0
F32_LOWEST = torch.finfo(torch.float32).min


def _interleave_blanks(targets):
    """[N, S] labels -> [N, 2S+1] lattice labels  b l1 b l2 ... lS b   (ha/ctc.py:127-129)."""
    N, S = targets.shape
    ext = targets.new_full((N, 2 * S + 1), BLANK)
    ext[:, 1::2] = targets
    return ext


def ctc_alpha_batch(emissions, targets, neg=F32_LOWEST):
    """Full log-alpha lattice [T, N, 2S+1] of ha/ctc.py:135-167 (ctc_forward_score3).

    ``neg`` is the value used for "log 0": the reference uses finfo(float32).min (ctc.py:135),
    not -inf.  The recursion runs over ALL 2S+1 states and ALL T frames regardless of the
    per-utterance lengths; lengths only select the read-out cell (ctc.py:169-174).
    """
    T, N, C = emissions.shape
    ext = _interleave_blanks(targets)
    S_ = ext.shape[1]
    alpha = emissions.new_full((T, N, S_), neg)
    rows = torch.arange(N)
    # skip transition s-2 -> s is allowed only into a non-blank that differs from l'[s-2]
    can_skip = torch.zeros(N, S_, dtype=torch.bool)
    if S_ > 2:
        can_skip[:, 2:] = (ext[:, 2:] != BLANK) & (ext[:, 2:] != ext[:, :-2])
    em0 = emissions[0]
    alpha[0, :, 0] = em0[rows, ext[:, 0]]
    if S_ > 1:
        alpha[0, :, 1] = em0[rows, ext[:, 1]]
    for t in range(1, T):
        prev = alpha[t - 1]
        stay = prev
        step = torch.cat([prev.new_full((N, 1), neg), prev[:, :-1]], dim=1)
        jump = torch.cat([prev.new_full((N, 2), neg), prev[:, :-2]], dim=1)
        base = torch.logaddexp(stay, step)
        base[:, 0] = prev[:, 0]                       # state 0 has a self loop only (ctc.py:148-149)
        with_skip = torch.logaddexp(base, jump)
        trans = torch.where(can_skip, with_skip, base)
        alpha[t] = trans + emissions[t].gather(1, ext)
    return alpha


def ctc_forward_score3(emissions, targets, emission_lengths, target_lengths):
    """Per-utterance negative log likelihood [N]  (ha/ctc.py:110-174)."""
    alpha = ctc_alpha_batch(emissions, targets)
    N = targets.shape[0]
    rows = torch.arange(N)
    t_last = emission_lengths.long() - 1
    s_last = 2 * target_lengths.long()
    final = alpha[t_last, rows]                       # [N, S_]
    a = final[rows, s_last]
    b = final[rows, s_last - 1]
    return -torch.logaddexp(a, b)


def ctc_reduce_mean(losses, target_lengths):
    """ha/ctc.py:177-178."""
    return (losses / target_lengths).mean(-1)


def _single_lattice(emissions, targets, wrap_skip_into_first):
    """Shared body of ctc_forward_score1 (ctc.py:4-50) and ctc_forward_score2 (ctc.py:54-107).

    Both single-utterance variants use -inf, never update state 0 after t=0 (the loops start
    at s=1 / write [t, 1:]), and read out the last two states at T-1.  score1 additionally
    reads ``log_alpha[t-1, s-2]`` with s=1, i.e. python index -1 = the LAST state
    (ctc.py:29,43): ``wrap_skip_into_first`` reproduces that.
    """
    T, C = emissions.shape
    S = targets.shape[0]
    ext = _interleave_blanks(targets[None])[0]
    S_ = 2 * S + 1
    ninf = float('-inf')
    alpha = emissions.new_full((T, S_), ninf)
    alpha[0, 0] = emissions[0, ext[0]]
    alpha[0, 1] = emissions[0, ext[1]]
    for t in range(1, T):
        for s in range(1, S_):
            stay = alpha[t - 1, s]
            step = alpha[t - 1, s - 1]
            acc = torch.logaddexp(stay, step)
            if s >= 2:
                if ext[s] != BLANK and ext[s] != ext[s - 2]:
                    acc = torch.logaddexp(acc, alpha[t - 1, s - 2])
            elif wrap_skip_into_first:
                # s == 1: ext[-1] is the trailing blank, ext[1] a label -> skip term is taken
                acc = torch.logaddexp(acc, alpha[t - 1, S_ - 1])
            alpha[t, s] = acc + emissions[t, ext[s]]
    return -torch.logaddexp(alpha[T - 1, S_ - 1], alpha[T - 1, S_ - 2]), alpha


def ctc_forward_score1(emissions, targets):
    return _single_lattice(emissions, targets, wrap_skip_into_first=True)[0]


def ctc_forward_score2(emissions, targets):
    return _single_lattice(emissions, targets, wrap_skip_into_first=False)[0]


def greedy_decode(log_probs):
    """Greedy CTC decode of ha/recognizer.py:48-59 (input lengths are ignored there too).

    Returns (list of python int lists, lengths [N] int64, alignments [N,T] int64, scores [N,T]).
    """
    scores, alignments = log_probs.max(dim=-1)
    hyps = []
    for row in alignments.tolist():
        out, prev = [], None
        for sym in row:
            if sym != prev and sym != BLANK:
                out.append(sym)
            prev = sym
        hyps.append(out)
    lengths = torch.tensor([len(h) for h in hyps], dtype=torch.int64)
    return hyps, lengths, alignments, scores


def _first_index(seqs, wanted):
    for i, q in enumerate(seqs):
        if q == wanted:
            return i
    return -1


def ctc_beam_search_decode_logits(emit_logits, beam_size=3, dtype=torch.float32):
    """Restatement of ha/beam.py:71-137, quirks included:

    * extension candidates enter with blank score 0.0 in the LOG domain (beam.py:125);
    * blank (k=0) is proposed as an output symbol (beam.py:123);
    * equal prefixes are never merged; the parent look-up takes the FIRST equal prefix and
      sees that parent's blank score already advanced to this frame iff parent index < s
      (beam.py:99-110);
    * candidate order = kept prefixes, then per prefix its V extensions (beam.py:93,123);
      ranking by torch.topk(sorted=True) (beam.py:129).
    """
    T, V = emit_logits.shape
    seqs = [[]]
    total = torch.zeros(1, dtype=dtype)
    blank = torch.zeros(1, dtype=dtype)
    label = torch.full((1,), float('-inf'), dtype=dtype)
    ks = torch.arange(V)
    for t in range(T):
        frame = emit_logits[t]
        nb = len(seqs)
        ext = torch.zeros(nb, V, dtype=dtype)
        for s in range(nb):
            q = seqs[s]
            if q:
                last = q[-1]
                label[s] = label[s] + frame[last]
                p = _first_index(seqs, q[:-1])
                if p >= 0:
                    label[s] = torch.logaddexp(label[s], frame[last] + 0 + blank[p])
            blank[s] = total[s] + frame[BLANK]
            pivot = q[-1] if q else BLANK
            base = torch.where(ks == pivot, blank[s], total[s])
            ext[s] = frame + 0. + base
        cands = seqs + [q + [k] for q in seqs for k in range(V)]
        blank_all = torch.cat([blank, torch.zeros(nb * V, dtype=dtype)])
        label_all = torch.cat([label, ext.reshape(-1)])
        total_all = torch.logaddexp(blank_all, label_all)
        top = total_all.topk(beam_size, dim=0, largest=True, sorted=True)
        total = top.values
        blank = blank_all[top.indices]
        label = label_all[top.indices]
        seqs = [cands[i] for i in top.indices.tolist()]
    return seqs, total


def ctc_beam_search_decode_probs(emit_probs, beam_size=3):
    """Probability-domain twin, ha/beam.py:5-68.

    The reference function raises NameError at beam.py:46 (``device`` is undefined in its
    scope).  This restatement computes what it computes once a module-global ``device`` exists
    (that is how tests/golden/make_golden.py runs it): same candidate order and the same
    quirks as the log-domain version, with extension blank-probability 0.0 (beam.py:56).
    """
    T, V = emit_probs.shape
    seqs = [[]]
    total = torch.ones(1, dtype=torch.float32)
    blank = torch.ones(1, dtype=torch.float32)
    label = torch.zeros(1, dtype=torch.float32)
    ks = torch.arange(V)
    for t in range(T):
        frame = emit_probs[t]
        nb = len(seqs)
        ext = torch.zeros(nb, V, dtype=torch.float32)
        for s in range(nb):
            q = seqs[s]
            if q:
                last = q[-1]
                label[s] = label[s] * frame[last]
                p = _first_index(seqs, q[:-1])
                if p >= 0:
                    label[s] = label[s] + frame[last] * 1 * blank[p]
            blank[s] = total[s] * frame[BLANK]
            pivot = q[-1] if q else BLANK
            onehot = (ks == pivot).float()
            base = onehot * blank[s] + (1 - onehot) * total[s]
            ext[s] = frame * 1. * base
        cands = seqs + [q + [k] for q in seqs for k in range(V)]
        blank_all = torch.cat([blank, torch.zeros(nb * V, dtype=torch.float32)])
        label_all = torch.cat([label, ext.reshape(-1)])
        total_all = blank_all + label_all
        top = total_all.topk(beam_size, dim=0, largest=True, sorted=True)
        total = top.values
        blank = blank_all[top.indices]
        label = label_all[top.indices]
        seqs = [cands[i] for i in top.indices.tolist()]
    return seqs, total


def ctc_brute_force_nll(log_probs, target):
    """-log sum over all alignments (exhaustive; T,V tiny).  Independent check of the lattice."""
    T, V = log_probs.shape
    target = list(target)
    total = -math.inf

    def collapse(path):
        out, prev = [], None
        for sym in path:
            if sym != prev and sym != BLANK:
                out.append(sym)
            prev = sym
        return out

    def rec(t, path, score):
        nonlocal total
        if t == T:
            if collapse(path) == target:
                m = max(total, score)
                total = m + math.log(math.exp(total - m) + math.exp(score - m)) if m > -math.inf else score
            return
        for k in range(V):
            rec(t + 1, path + [k], score + float(log_probs[t, k]))

    rec(0, [], 0.0)
    return -total
