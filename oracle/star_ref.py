"""CPU restatement of the reference's two further lattices: star-CTC (ha/star.py) and the transducer forward score
(ha/transducer.py) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Written state by state / cell by cell (explicit Python loops, torch float32 scalars on the CPU, so every logaddexp is the ATen one
the reference runs) instead of the reference's shifted-slice and parallel-scan formulations: the restatement shows the lattice the
HIP kernels (haloop_amd/csrc/lattice.hip) walk.  Pinned against reference-generated fixtures (tests/golden/g8_star.npz,
g9_transducer.npz; tests/golden/make_golden.py imports ha.star / ha.transducer) in tests/test_oracle_golden.py.  The gradients are
the analytic alpha-beta ones, checked there against the reference's autograd gradients.
"""
import torch

VOID = torch.finfo(torch.float32).min            # the reference's "log 0" (ha/star.py:91, ha/transducer.py:187)


def _lae(a, b):
    return torch.logaddexp(a, b)


# ------------------------------------------------------------------------------------------------ star-CTC
def star_emissions(log_probs, penalty=0.0):
    """ha/star.py:9-41 (intersperse_stars, emission half): [T, N, V] -> [T, N, 2V]: the V symbols, then <star> = logsumexp of all
    non-blank symbols at index V, then for every symbol s >= 1 at index V + s "<star> without s" = logsubexp(<star>, s).
    star_ctc_forward_score calls it with the DEFAULT penalty 0 (ha/star.py:82), whatever its own star_penalty is."""
    complete = log_probs[:, :, 1:].logsumexp(dim=-1, keepdim=True)
    starsub = complete + torch.log1p(-torch.exp(log_probs[:, :, 1:] - complete))
    return torch.cat([log_probs, complete + penalty, starsub + penalty], dim=-1)


def star_states(targets, V):
    """The 4S+3 lattice symbols of one batch (ha/star.py:46-49, 53-66): for labels a b c
    blank <star\\a> blank a blank <star\\b> blank b blank <star\\c> blank c blank <star> blank, as ids into the 2V-wide emissions
    (<star\\s> = V + s; a label 0, i.e. target padding, therefore gets V + 0 = the complete <star>)."""
    N, S = targets.shape
    ids = targets.new_zeros((N, 4 * S + 3))
    ids[:, 1:4 * S:4] = V + targets
    ids[:, 3:4 * S:4] = targets
    ids[:, 4 * S + 1] = V
    return ids


def star_ctc_forward_score(emissions, targets, emission_lengths, target_lengths, star_penalty=-0.5, return_alpha=False):
    """ha/star.py:65-166.  State i of the 4S+3 states is a blank (i even), a star (i % 4 == 1) or a label (i % 4 == 3).  Into state i
    at frame t from frame t-1 (ha/star.py:112-135):
        blank:  i-1, i
        star:   i-1, i, i+1 (the blank AFTER it), plus star_penalty; state 4S+3 (past the end) reads the constant -7007.7007 (:101)
        label:  i-3, i-1, i-2, and i-4 (the previous label) unless both labels are equal; NO self loop.
    Four virtual states before state 0 hold 0 at frame 0 (:93), so frame 1 can start in state 0 (from i-1) or state 3 (from i-4).
    The recursion runs over all T frames and all states; lengths only pick the read-out: frame emission_lengths, the last four states
    of the first 4*target_lengths+3 (:153-162)."""
    T, N, C = emissions.shape
    S = targets.shape[1]
    em = star_emissions(emissions)
    ids = star_states(targets, C)
    S_ = 4 * S + 3
    same = torch.zeros(N, S_, dtype=torch.bool)
    for k in range(1, S):
        same[:, 4 * k + 3] = targets[:, k] == targets[:, k - 1]
    toot = torch.tensor(-7007.7007)
    void = torch.tensor(VOID)
    pen = torch.tensor(float(star_penalty))
    alpha = emissions.new_full((T + 1, N, S_), VOID)      # alpha[t]: after frame t (1-based); alpha[0]: the real states at frame 0 (void)
    for n in range(N):
        for t in range(1, T + 1):
            def prev(i):
                if i < 0:
                    return torch.tensor(0.0) if t == 1 else void
                if i >= S_:
                    return toot
                return alpha[t - 1, n, i]
            for i in range(S_):
                if i % 2 == 0:
                    tr = _lae(prev(i - 1), prev(i))
                elif i % 4 == 1:
                    tr = _lae(_lae(prev(i - 1), prev(i)), prev(i + 1)) + pen
                else:
                    tr = _lae(_lae(prev(i - 3), prev(i - 1)), prev(i - 2))
                    if not same[n, i]:
                        tr = _lae(tr, prev(i - 4))
                alpha[t, n, i] = tr + em[t - 1, n, ids[n, i]]
    losses = []
    for n in range(N):
        t_last, s_last = int(emission_lengths[n]), 4 * int(target_lengths[n]) + 2
        a = alpha[t_last, n]
        losses.append(-_lae(_lae(_lae(a[s_last], a[s_last - 1]), a[s_last - 2]), a[s_last - 3]))
    losses = torch.stack(losses)
    return (losses, alpha) if return_alpha else losses


def star_ctc_grad(emissions, targets, emission_lengths, target_lengths, star_penalty=-0.5):
    """d sum(losses) / d emissions [T, N, C] by alpha-beta: -occupancy of every state, carried from the 2C-wide star emissions back to the
    C symbols (d <star> / d lp[k] = exp(lp[k] - <star>) for k >= 1; d <star\\s> / d lp[k] = exp(lp[k] - <star\\s>) for k >= 1, k != s)."""
    T, N, C = emissions.shape
    S = targets.shape[1]
    S_ = 4 * S + 3
    em = star_emissions(emissions.double())
    ids = star_states(targets, C)
    losses, alpha = star_ctc_forward_score(emissions, targets, emission_lengths, target_lengths, star_penalty, return_alpha=True)
    alpha = alpha.double()
    ninf = torch.tensor(float('-inf'), dtype=torch.float64)
    grad = torch.zeros(T, N, C, dtype=torch.float64)
    for n in range(N):
        Tn, s_last = int(emission_lengths[n]), 4 * int(target_lengths[n]) + 2
        logz = -losses[n].double()
        same = [False] * S_
        for k in range(1, S):
            same[4 * k + 3] = bool(targets[n, k] == targets[n, k - 1])
        beta = torch.full((S_,), float('-inf'), dtype=torch.float64)
        beta[s_last - 3:s_last + 1] = 0.0
        for t in range(Tn, 0, -1):
            occ = torch.exp(alpha[t, n] + beta - logz)               # posterior of being in state i at frame t
            g_star = torch.zeros(2 * C, dtype=torch.float64)
            g_star.index_add_(0, ids[n], -occ)
            lp = emissions[t - 1, n].double()
            g = g_star[:C].clone()
            g[1:] += g_star[C] * torch.exp(lp[1:] - em[t - 1, n, C])
            for s in range(1, C):
                if g_star[C + s] != 0:
                    term = g_star[C + s] * torch.exp(lp[1:] - em[t - 1, n, C + s])
                    term[s - 1] = 0.0
                    g[1:] += term
            grad[t - 1, n] = g
            # beta of frame t-1: state j reaches i = j, j+1 (blank / star), j-1 (star from the blank after it), and labels i = j+1..j+4
            nb = torch.full((S_,), float('-inf'), dtype=torch.float64)
            e = em[t - 1, n, ids[n]] + beta                              # emission + beta of the frame-t state
            for i in range(S_):
                if e[i] == ninf:
                    continue
                if i % 2 == 0:
                    srcs = [(i - 1, 0.0), (i, 0.0)]
                elif i % 4 == 1:
                    srcs = [(i - 1, star_penalty), (i, star_penalty), (i + 1, star_penalty)]
                else:
                    srcs = [(i - 3, 0.0), (i - 1, 0.0), (i - 2, 0.0)] + ([] if same[i] else [(i - 4, 0.0)])
                for j, w in srcs:
                    if 0 <= j < S_:
                        nb[j] = torch.logaddexp(nb[j], e[i] + w)
            beta = nb
    return grad.float()


# --------------------------------------------------------------------------------------------- transducer
def transducer_forward_score(joint, targets, joint_lengths, target_lengths, return_alpha=False):
    """ha/transducer.py:175-207 cell by cell: joint [N, T, U+1, K] log-probabilities, blank 0.
        alpha[t, 0] = sum_{t' < t} joint[t', 0, 0]                                              (:189-192)
        alpha[t, u] = logaddexp(alpha[t, u-1] + joint[t, u-1, y[u-1]], alpha[t-1, u] + joint[t-1, u, 0])   (:197-203; t = 0: first term)
        loss = -(alpha[T_n - 1, U_n] + joint[T_n - 1, U_n, 0])                                  (:205-206)
    over ALL T and U+1 whatever the lengths.  The reference evaluates the recurrence over t with a log-space parallel scan
    (ha/scan.py:93-126), i.e. the same sums in another association: results agree to float32 rounding (1e-5 relative here)."""
    N, T, U1, K = joint.shape
    alpha = joint.new_full((N, T, U1), VOID)
    for n in range(N):
        acc = torch.tensor(0.0)
        for t in range(T):
            alpha[n, t, 0] = acc
            acc = acc + joint[n, t, 0, 0]
        for u in range(1, U1):
            y = int(targets[n, u - 1])
            for t in range(T):
                bot = alpha[n, t, u - 1] + joint[n, t, u - 1, y]
                alpha[n, t, u] = bot if t == 0 else _lae(bot, alpha[n, t - 1, u] + joint[n, t - 1, u, 0])
    rows = torch.arange(N)
    tl, ul = joint_lengths.long() - 1, target_lengths.long()
    losses = -(alpha[rows, tl, ul] + joint[rows, tl, ul, 0])
    return (losses, alpha) if return_alpha else losses


def transducer_grad(joint, targets, joint_lengths, target_lengths):
    """d sum(losses) / d joint by alpha-beta: only joint[n, t, u, 0] and joint[n, t, u, y[u]] of cells inside the [T_n, U_n + 1] lattice
    carry gradient."""
    N, T, U1, K = joint.shape
    losses, alpha = transducer_forward_score(joint, targets, joint_lengths, target_lengths, return_alpha=True)
    alpha, jd = alpha.double(), joint.double()
    grad = torch.zeros(N, T, U1, K, dtype=torch.float64)
    for n in range(N):
        Tn, Un = int(joint_lengths[n]), int(target_lengths[n])
        logz = -losses[n].double()
        beta = torch.full((Tn + 1, Un + 2), float('-inf'), dtype=torch.float64)
        for t in range(Tn - 1, -1, -1):
            for u in range(Un, -1, -1):
                if t == Tn - 1 and u == Un:
                    beta[t, u] = jd[n, t, u, 0]
                    continue
                b = beta[t + 1, u] + jd[n, t, u, 0]
                if u < Un:
                    b = torch.logaddexp(b, beta[t, u + 1] + jd[n, t, u, int(targets[n, u])])
                beta[t, u] = b
        for t in range(Tn):
            for u in range(Un + 1):
                nxt = torch.tensor(0.0, dtype=torch.float64) if (t == Tn - 1 and u == Un) else beta[t + 1, u]
                grad[n, t, u, 0] -= torch.exp(alpha[n, t, u] + jd[n, t, u, 0] + nxt - logz)
                if u < Un:
                    y = int(targets[n, u])
                    grad[n, t, u, y] -= torch.exp(alpha[n, t, u] + jd[n, t, u, y] + beta[t, u + 1] - logz)
    return grad.float()
