"""The fused CTC head (csrc/head.hip: three launches) against the separate operators it replaces and against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _separate(ops, lib, feats, W, b, drop, il, tg, tl):
    B, T, H = feats.shape
    V = W.shape[0]
    fdrop = ops.dropout_fwd(feats, drop, lib.HALO_STREAM_CLASSIFIER) if drop.p > 0 else feats
    f2d = fdrop.view(B * T, H)
    logits = ops.gemm(f2d, W, True, True, B * T, V, H, bias1=b)
    lp = ops.log_softmax_fwd(logits)
    flen, grad_out = ops.ctc_prepare(il, tl)
    nll, alpha, saved = ops.ctc_fwd(lp.view(B, T, V), False, tg, flen, tl)
    loss = torch.zeros((), device=DEV)
    ops.ctc_mean_loss(nll, tl, loss)
    dlp = ops.ctc_bwd(lp.view(B, T, V), False, saved, alpha, nll, grad_out)
    dlogits = ops.log_softmax_bwd(dlp.view(B * T, V), lp)
    dW = ops.gemm(dlogits, f2d, False, False, V, H, B * T)
    db = ops.colsum(dlogits)
    dfeats = ops.gemm(dlogits, W, True, False, B * T, H, V, drop=drop, stream_id=lib.HALO_STREAM_CLASSIFIER)
    return dict(lp=lp.view(B, T, V), nll=nll, flen=flen, loss=loss, dW=dW, db=db, dfeats=dfeats.view(B, T, H), alpha=alpha)


@pytest.mark.parametrize('B,T_in,H,V,S,p', [(64, 80, 1024, 32, 10, 0.2), (3, 41, 64, 9, 4, 0.0), (5, 120, 256, 32, 12, 0.3), (1, 9, 128, 5, 1, 0.0)])
def test_fused_head_equals_separate_operators(B, T_in, H, V, S, p):
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch()
    T = (T_in + 6 - 5) // 4 + 1
    assert ops.ctc_head_supported(T, H, V, S)
    g = torch.Generator().manual_seed(B * 7 + H)
    feats = torch.randn(B, T, H, generator=g).relu().to(DEV)
    W = (torch.randn(V, H, generator=g) / H ** 0.5).to(DEV)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV)
    il = torch.tensor([T_in - 3 * (i % 5) for i in range(B)], dtype=torch.int64)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
    if B >= 3:
        il[1] = 9; tg[1] = 1; tl[1] = min(S, 4)          # infeasible when S >= 4 (equal labels need 2S-1 frames): nll = inf
        tl[2] = 0                                        # an empty target: only blanks
    il, tg, tl = il.to(DEV), tg.to(DEV), tl.to(DEV)
    drop = ops.Dropout(p, 0x1234ABCD5678, 5) if p > 0 else ops.NO_DROPOUT
    want = _separate(ops, _lib, feats, W, b, drop, il, tg, tl)
    loss = torch.zeros((), device=DEV)
    ticket = torch.zeros(1, device=DEV, dtype=torch.int32)
    dW, db = torch.empty_like(W), torch.empty_like(b)
    for rep in range(2):                                  # twice: the ticket must come back to zero
        lp, alpha, nll, flen, grad_out, (tg64, tl64) = ops.ctc_head_fwd(feats, W, b, drop, _lib.HALO_STREAM_CLASSIFIER, il, tg, tl, loss, ticket)
        dfeats = ops.ctc_head_bwd(feats, W, drop, _lib.HALO_STREAM_CLASSIFIER, flen, tg64, tl64, lp, alpha, nll, grad_out, dW, db)
        assert int(ticket.item()) == 0
        assert torch.equal(flen.cpu(), want['flen'].cpu())
        np.testing.assert_allclose(lp.cpu().numpy(), want['lp'].cpu().numpy(), atol=2e-6)
        fin = torch.isfinite(want['nll']).cpu().numpy()
        assert np.array_equal(np.isfinite(nll.cpu().numpy()), fin)
        np.testing.assert_allclose(nll.cpu().numpy()[fin], want['nll'].cpu().numpy()[fin], rtol=2e-6, atol=2e-6)
        if fin.all():
            np.testing.assert_allclose(loss.item(), want['loss'].item(), rtol=2e-6)
            for k, got in (('dW', dW), ('db', db), ('dfeats', dfeats)):
                ref = want[k].cpu().numpy()
                np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=2e-4, atol=1e-6 + 2e-5 * np.abs(ref).max(), err_msg=k)
        else:
            assert not np.isfinite(loss.item())


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
@pytest.mark.parametrize('B,T_in,H,V,S,p', [(64, 80, 1024, 32, 10, 0.2), (3, 41, 64, 9, 4, 0.0), (5, 120, 256, 32, 12, 0.3), (1, 9, 128, 5, 1, 0.0),
                                            (128, 80, 1024, 32, 10, 0.2), (256, 80, 1024, 32, 10, 0.2), (1, 80, 1024, 32, 10, 0.2),
                                            (40, 60, 1536, 29, 7, 0.1)])
def test_head_in_one_launch_equals_the_two_launches(B, T_in, H, V, S, p, mode):
    """halo_ctc_head_train (forward + backward of the head in one launch: slices of an utterance exchanging partial logits, split-bf16
    products, alpha and beta side by side) against the two exact-f32 launches it replaces, at fp32-grade tolerances: 8 / 4 / 2 / 1
    slices per utterance, ragged lengths, an infeasible and an empty target, with and without classifier dropout; run three times
    over the same ticket words (each launch's partial sums carry its own tag)."""
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch()
    T = (T_in + 6 - 5) // 4 + 1
    assert ops.ctc_head_supported(T, H, V, S)
    g = torch.Generator().manual_seed(B * 7 + H)
    feats = torch.randn(B, T, H, generator=g).relu().to(DEV)
    W = (torch.randn(V, H, generator=g) / H ** 0.5).to(DEV)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV)
    il = torch.tensor([T_in - 3 * (i % 5) for i in range(B)], dtype=torch.int64)
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(max(1, S // 2), S + 1, (B,), generator=g)
    if B >= 3:
        il[1] = 9; tg[1] = 1; tl[1] = min(S, 4)
        tl[2] = 0
    il, tg, tl = il.to(DEV), tg.to(DEV), tl.to(DEV)
    drop = ops.Dropout(p, 0x1234ABCD5678, 5) if p > 0 else ops.NO_DROPOUT
    sid = _lib.HALO_STREAM_CLASSIFIER
    loss0 = torch.zeros((), device=DEV)
    t0 = torch.zeros(1, device=DEV, dtype=torch.int32)
    dW0, db0 = torch.empty_like(W), torch.empty_like(b)
    lp0, alpha0, nll0, flen0, go0, (tg64, tl64) = ops.ctc_head_fwd(feats, W, b, drop, sid, il, tg, tl, loss0, t0)
    dfeats0 = ops.ctc_head_bwd(feats, W, drop, sid, flen0, tg64, tl64, lp0, alpha0, nll0, go0, dW0, db0)
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        loss = torch.full((), -1.0, device=DEV)
        ticket = ops.ctc_head_train_ticket(B, H, DEV)
        dW, db = torch.empty_like(W), torch.empty_like(b)
        for rep in range(3):
            dfeats, nll, flen, lp = ops.ctc_head_train(feats, W, b, drop, sid, il, tg, tl, loss, ticket, dW, db, want_lp=True)
            assert ticket[:2].tolist() == [0, rep + 1]                   # the loss ticket back at zero, the launches counted
            assert torch.equal(flen, flen0)
            np.testing.assert_allclose(lp.cpu().numpy(), lp0.cpu().numpy(), atol=3e-5)
            fin = torch.isfinite(nll0).cpu().numpy()
            assert np.array_equal(np.isfinite(nll.cpu().numpy()), fin)
            np.testing.assert_allclose(nll.cpu().numpy()[fin], nll0.cpu().numpy()[fin], rtol=2e-5, atol=2e-5)
            if fin.all():
                np.testing.assert_allclose(loss.item(), loss0.item(), rtol=2e-5)
                for k, got, ref in (('dW', dW, dW0), ('db', db, db0), ('dfeats', dfeats, dfeats0)):
                    ref = ref.cpu().numpy()
                    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=3e-4, atol=1e-6 + 3e-5 * np.abs(ref).max(), err_msg=k)
            else:
                assert not np.isfinite(loss.item())
                ok = fin[:, None, None] & np.ones((1, T, H), dtype=bool)
                np.testing.assert_allclose(dfeats.cpu().numpy()[ok], dfeats0.cpu().numpy()[ok], rtol=3e-4,
                                           atol=1e-6 + 3e-5 * np.abs(dfeats0.cpu().numpy()[ok]).max())
        _lib.set_math_mode('f32')
        with pytest.raises(Exception):                       # the exact-f32 mode keeps its exact-f32 launches
            ops.ctc_head_train(feats, W, b, drop, sid, il, tg, tl, loss, ticket, dW, db)
    finally:
        _lib.set_math_mode(prev)


def test_a_slice_that_never_publishes_is_loud_and_the_next_launch_is_clean():
    """The one-launch head with one slice of one utterance made mute by the test hook (halo_debug_mute_workgroup): its peers' bounded
    waits give up, the caller's sticky status word is raised, the loss is not finite -- and the launch still ends with its loss ticket
    back at zero and its launch count advanced, so the next launch (unmuted) matches none of the stale pairs and gives the clean
    result bit for bit."""
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch()
    B, T_in, H, V, S = 16, 80, 1024, 32, 10
    T = (T_in + 6 - 5) // 4 + 1
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(B, T, H, generator=g).relu().to(DEV)
    W = (torch.randn(V, H, generator=g) / H ** 0.5).to(DEV)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV)
    il = torch.full((B,), T_in, dtype=torch.int64).to(DEV)
    tg = torch.randint(1, V, (B, S), generator=g).to(DEV)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g).to(DEV)
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    _lib.set_status_word(status)
    try:
        loss = torch.zeros((), device=DEV)
        ticket = ops.ctc_head_train_ticket(B, H, DEV)
        assert ticket.numel() > 2                                  # more than one slice per utterance: there is an exchange to break
        dW, db = torch.empty_like(W), torch.empty_like(b)
        sid = _lib.HALO_STREAM_CLASSIFIER
        dfeats, nll, _, _ = ops.ctc_head_train(feats, W, b, ops.NO_DROPOUT, sid, il, tg, tl, loss, ticket, dW, db)
        clean = (loss.clone(), dfeats.clone(), nll.clone(), dW.clone())
        assert int(status.item()) == 0 and ticket[:2].tolist() == [0, 1]
        _lib.check(_lib.lib().halo_debug_mute_workgroup(5), 'mute')
        ops.ctc_head_train(feats, W, b, ops.NO_DROPOUT, sid, il, tg, tl, loss, ticket, dW, db)
        assert int(status.item()) != 0 and not np.isfinite(loss.item())
        assert ticket[:2].tolist() == [0, 2]                       # every utterance took its ticket; the count advanced
        _lib.check(_lib.lib().halo_debug_mute_workgroup(-1), 'unmute')
        status.zero_()
        dfeats, nll, _, _ = ops.ctc_head_train(feats, W, b, ops.NO_DROPOUT, sid, il, tg, tl, loss, ticket, dW, db)
        assert int(status.item()) == 0 and ticket[:2].tolist() == [0, 3]
        assert torch.equal(loss, clean[0]) and torch.equal(dfeats, clean[1]) and torch.equal(nll, clean[2]) and torch.equal(dW, clean[3])
    finally:
        _lib.lib().halo_debug_mute_workgroup(-1)
        _lib.set_status_word(None)
        _lib.set_math_mode(prev)


def test_fused_head_refuses_unsupported_shapes():
    from haloop_amd import ops
    assert not ops.ctc_head_supported(40, 1024, 32, 10)      # more than 32 frames
    assert not ops.ctc_head_supported(21, 1024, 256, 10)     # more than 32 classes
    assert not ops.ctc_head_supported(21, 96, 32, 10)        # H % 64 != 0
    assert not ops.ctc_head_supported(21, 1024, 32, 40)      # 2S+1 > 64


@pytest.mark.parametrize('fused', [True, False])
def test_trainer_step_with_and_without_fused_head_match_oracle(fused):
    """One LC-style training step (dropout off) vs the CPU oracle's Trainer, on the fused head and on the separate operators."""
    from haloop_amd import _lib, rnn, recognizer
    from haloop_amd.train import LstmCtcTrainer
    from oracle import cpu_ref
    _lib.lib(); _lib.set_math_mode('bf16x3')
    F_, C, H, L, V, B, T, S = 20, 32, 64, 2, 11, 6, 45, 5
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 3)
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 4)
    il = torch.clamp(il - torch.arange(B) % 4, min=T // 2)
    enc = rnn.Encoder(F_, C, H, num_layers=L); rec = recognizer.TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=False, fused_head=fused)
    ref = cpu_ref.Trainer(enc_p, rec_p, lr=3e-3)
    for _ in range(2):
        loss = tr.step(x.to(DEV), il.to(DEV), tg.to(DEV), tl.to(DEV))
        loss_ref, gn_ref = ref.step(x, il, tg, tl)
        np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-5)
        np.testing.assert_allclose(tr.grad_norm.item(), gn_ref.item(), rtol=2e-4)
    for k, v in enc.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), ref.enc[k].detach().numpy(), atol=3e-5, err_msg=k)
    for k, v in rec.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), ref.rec[k].detach().numpy(), atol=3e-5, err_msg=k)


@pytest.mark.parametrize('B,T,H,V', [(64, 21, 1024, 32), (5, 32, 512, 29), (3, 1, 64, 2), (130, 7, 1536, 32)])
def test_head_greedy_matches_the_separate_launches(B, T, H, V):
    """classifier + log_softmax + greedy collapse in one launch (TemporalClassifier.decode, ha/recognizer.py:48-59) against the three
    launches it replaces in the inference path (product, log-softmax, ctc_greedy) and the oracle's greedy on the same log-probs."""
    from haloop_amd import ops
    from oracle import lattice
    g = torch.Generator().manual_seed(B * 7 + T)
    feats = torch.randn(B, T, H, generator=g).cuda()
    W = (torch.randn(V, H, generator=g) * H ** -0.5).cuda()
    W[0] += 0.02 * feats.mean((0, 1))            # some blanks among the winners
    b = (0.1 * torch.randn(V, generator=g)).cuda()
    ali, scores, hyp, hyp_len, lp = ops.ctc_head_greedy(feats, W, b, want_lp=True)
    logits = ops.gemm(feats.view(B * T, H), W, True, True, B * T, V, H, bias1=b)
    lp2 = ops.log_softmax_fwd(logits).view(B, T, V)
    torch.testing.assert_close(lp, lp2, rtol=0, atol=5e-5)            # (both products split-bf16 outside the exact-f32 mode: fp32-grade)
    ali2, scores2, hyp2, len2 = ops.ctc_greedy(lp)            # the same log-probs: index results are exact
    assert torch.equal(ali, ali2) and torch.equal(hyp, hyp2) and torch.equal(hyp_len, len2)
    assert torch.equal(scores, scores2)
    # the run without the log-prob output writes the same results
    out = ops.ctc_head_greedy(feats, W, b)
    assert torch.equal(out[0], ali) and torch.equal(out[2], hyp) and torch.equal(out[3], hyp_len) and torch.equal(out[1], scores)
    hyps, lengths, alignments, best = lattice.greedy_decode(lp.cpu())
    assert torch.equal(ali.cpu(), alignments) and torch.equal(hyp_len.cpu(), lengths) and torch.equal(scores.cpu(), best)
    for n in range(B):
        assert hyp[n, :lengths[n]].tolist() == hyps[n] and not hyp[n, lengths[n]:].any()


def test_deferred_small_reductions_ride_in_the_conv_backward_launch():
    """ops.defer_small_jobs(): the head's partial sums and the two-layer LSTM launch's bias-gradient sums are queued and run from the tail
    blocks of the conv backward's reduce launch (csrc/small_jobs.h) -- bitwise what their own launches give; a block that ends with jobs
    still queued flushes them in one launch; nothing stays queued or switched on afterwards."""
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch()
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        B, T, C, H, V, S, L, F_, ks = 64, 20, 128, 256, 32, 8, 2, 16, 5
        g = torch.Generator().manual_seed(3)
        k = 1.0 / H ** 0.5
        x = torch.relu(torch.randn(T, B, C, generator=g)).to(DEV)
        col = torch.randn(T * B, F_ * ks, generator=g).to(DEV)
        w_ih = [((torch.rand(4 * H, C if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
        w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
        b_ih = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
        b_hh = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
        W = (torch.randn(V, H, generator=g) / H ** 0.5).to(DEV)
        b = (torch.randn(V, generator=g) * 0.1).to(DEV)
        T_in = 4 * (T - 1) + ks - 6
        il = torch.tensor([T_in - 3 * (i % 5) for i in range(B)], dtype=torch.int64).to(DEV)
        tg = torch.randint(1, V, (B, S), generator=g).to(DEV)
        tl = torch.randint(S // 2, S + 1, (B,), generator=g).to(DEV)
        assert _lib.lib().halo_lstm_persistent2_eligible(T, B, H, L)

        def run(defer, flush_by_conv=True):
            loss = torch.zeros((), device=DEV)
            ticket = torch.zeros(1, device=DEV, dtype=torch.int32)
            feats = torch.empty(B, T, H, device=DEV)
            _, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, y=feats, y_strides=(H, T * H), y_relu=True)
            dW, db = torch.full_like(W, float('nan')), torch.full_like(b, float('nan'))
            hws = ops.ctc_head_workspace(B, H, V, DEV)
            ws = ops.lstm_bwd_workspace(x, w_hh)
            if defer:
                ops.defer_small_jobs.begin()
            lp, alpha, nll, flen, grad_out, (tg64, tl64) = ops.ctc_head_fwd(feats, W, b, ops.NO_DROPOUT, _lib.HALO_STREAM_CLASSIFIER, il, tg, tl, loss, ticket)
            dfeats = ops.ctc_head_bwd(feats, W, ops.NO_DROPOUT, _lib.HALO_STREAM_CLASSIFIER, flen, tg64, tl64, lp, alpha, nll, grad_out, dW, db, workspace=hws)
            dx, grads = ops.lstm_bwd(x, w_ih, w_hh, dfeats.view(B * T, H), (H, T * H), True, reserve, workspace=ws, want_dx=True)
            if defer:
                torch.cuda.synchronize()
                assert torch.isnan(dW).all()                 # still queued: nothing has written the sums
            out = {}
            if flush_by_conv:
                out['cw'], out['cb'] = ops.subsample_bwd(dx, x, col, B, T_in, F_, C, 0.0)
            if defer:
                ops.defer_small_jobs.end()
            out.update(dW=dW, db=db, dfeats=dfeats, dx=dx)
            for name, lst in grads.items():
                for l, t in enumerate(lst):
                    out[f'{name}{l}'] = t
            torch.cuda.synchronize()
            return {k_: v.clone() for k_, v in out.items()}

        plain = run(False)
        for flush_by_conv in (True, False):
            got = run(True, flush_by_conv)
            for name, v in got.items():
                assert torch.equal(v, plain[name]), name
        assert _lib.lib().halo_flush_small_jobs(None) == 0
        again = run(False)                                   # the switch is off again: the head's own launch runs
        assert torch.equal(again['dW'], plain['dW'])
    finally:
        _lib.set_math_mode(prev)


def test_squared_gradient_norm_from_the_producing_launches():
    """ops.collect_grad_sumsq: the two-layer LSTM backward's weight-gradient launches, its (deferred) bias sums and the conv backward's
    reduce launch leave the squared-norm partials of what they store; their sum is the squared norm of exactly those gradients."""
    from haloop_amd import _lib, ops
    _lib.lib(); _lib.lend_scratch()
    prev = _lib.get_math_mode()
    _lib.set_math_mode('bf16')
    try:
        for (T, B, C, H) in ((20, 64, 128, 1024), (7, 32, 128, 256)):
            L, F_, ks = 2, 16, 5
            g = torch.Generator().manual_seed(T)
            k = 1.0 / H ** 0.5
            x = torch.relu(torch.randn(T, B, C, generator=g)).to(DEV)
            col = torch.randn(T * B, F_ * ks, generator=g).to(DEV)
            w_ih = [((torch.rand(4 * H, C if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
            w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
            b_ih = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
            b_hh = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
            dy = torch.randn(T, B, H, generator=g).to(DEV)
            T_in = 4 * (T - 1) + ks - 6
            _, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh)
            ws = ops.lstm_bwd_workspace(x, w_hh)
            parts = torch.full((4096,), float('nan'), device=DEV)
            ops.defer_small_jobs.begin()
            ops.collect_grad_sumsq(parts)
            dx, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, workspace=ws, want_dx=True)
            n1, bits1 = ops.grad_sumsq_state()
            assert bits1 == 15 and n1 > 0
            cw, cb = ops.subsample_bwd(dx, x, col, B, T_in, F_, C, 0.0)
            n, bits = ops.grad_sumsq_state()
            ops.collect_grad_sumsq(None)
            ops.defer_small_jobs.end()
            assert bits == ops.GRAD_SUMSQ_ALL and n1 < n <= parts.numel()
            assert ops.grad_sumsq_state() == (0, 0)
            got = parts[:n].double().sum().item()
            assert torch.isnan(parts[n:]).all()
            want = sum(t.double().pow(2).sum().item() for lst in grads.values() for t in lst) + cw.double().pow(2).sum().item() + cb.double().pow(2).sum().item()
            assert abs(got - want) <= 1e-5 * want, (got, want)
    finally:
        _lib.set_math_mode(prev)
