"""Size-independent properties of the LSTM-CTC path at BASELINE config 2's full size (LC-2x1024, B=64, 80 frames x 80 mels, vocab 32), through
the module API, eval mode (no dropout): what must hold whatever the numbers are.  The fixtures of tests/golden pin the values at this size
(test_gpu_lstm_b64.py); these pin the structure -- which batch row, which frame, which class an output may depend on."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
F_, C, H, L, V, B, T, S = 80, 128, 1024, 2, 32, 64, 80, 10


@pytest.fixture(scope='module')
def model():
    from haloop_amd import _lib, rnn, recognizer, synth
    _lib.lib(); _lib.lend_scratch()
    enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
    enc = rnn.Encoder(F_, C, H, num_layers=L); rec = recognizer.TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    x, il, tg, tl = synth.synthetic_batch(B, T, F_, V, S, 7)
    il = torch.tensor([T - 3 * (i % 7) for i in range(B)], dtype=torch.int64)        # ragged lengths
    return _lib, enc, rec, (x.to(DEV), il, tg, tl)


@pytest.fixture(params=['bf16', 'bf16x3'])
def math_mode(request, model):
    lib = model[0]
    prev = lib.get_math_mode()
    lib.set_math_mode(request.param)
    yield request.param
    lib.set_math_mode(prev)


def _nll(rec, feats, tg, flen, tl):
    """per-utterance negative log-likelihoods through the product's CTC"""
    from haloop_amd import ops
    lp = rec.log_probs(feats).detach().contiguous()
    nll, _, _ = ops.ctc_fwd(lp, False, tg.to(DEV), flen.to(DEV).long(), tl.to(DEV))
    return nll


def test_batch_rows_are_independent(model, math_mode):
    """Permuting the utterances permutes features, per-utterance losses and greedy hypotheses -- bit for bit: no output of one batch row
    depends on which tile, workgroup or lane the row sits in, nor on its neighbours."""
    _, enc, rec, (x, il, tg, tl) = model
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        f1, l1, _ = enc(x, il)
        f2, l2, _ = enc(x[perm.to(DEV)].contiguous(), il[perm])
        assert torch.equal(l1[perm], l2)
        assert torch.equal(f1[perm.to(DEV)], f2)
        n1, n2 = _nll(rec, f1, tg, l1, tl), _nll(rec, f2, tg[perm], l2, tl[perm])
        assert torch.equal(n1[perm.to(DEV)], n2)
        h1, hl1, a1, s1, _ = rec.decode(f1, il, tl)
        h2, hl2, a2, s2, _ = rec.decode(f2, il[perm], tl[perm])
        assert torch.equal(hl1[perm], hl2) and torch.equal(a1[perm.to(DEV)], a2) and torch.equal(s1[perm.to(DEV)], s2)
        for i, j in enumerate(perm.tolist()):
            assert torch.equal(h1[j], h2[i])
            assert (h1[j] != 0).all()                       # no blank survives the collapse


def test_features_are_causal(model, math_mode):
    """A unidirectional encoder: feature frame t' reads input frames up to 4 t' + 1 (kernel 5, stride 4, padding 3) and nothing later.
    Replacing the input from frame 42 on leaves the first ten feature frames of every utterance bit-identical and changes the rest."""
    _, enc, rec, (x, il, tg, tl) = model
    cut = 42
    x2 = x.clone()
    x2[:, cut:] = torch.randn(B, T - cut, F_, generator=torch.Generator().manual_seed(3)).to(DEV)
    with torch.no_grad():
        f1, _, _ = enc(x, il)
        f2, _, _ = enc(x2, il)
    last_clean = (cut - 2) // 4              # 4 t' + 1 < cut
    assert torch.equal(f1[:, :last_clean + 1], f2[:, :last_clean + 1])
    assert not torch.equal(f1[:, last_clean + 1:], f2[:, last_clean + 1:])


def test_mean_loss_and_gradients_of_a_doubled_batch(model, math_mode):
    """reduction='mean' over utterances: the batch of 32 and the same 32 twice (B = 64) have the same loss and the same parameter
    gradients (each copy contributes half), up to the order of the fp32 sums over the batch; and the gradient at the logits sums to zero
    over the classes in every frame (log-softmax), so the classifier's bias gradient sums to zero."""
    _, enc, rec, (x, il, tg, tl) = model
    h = B // 2
    out = []
    for xs, ils, tgs, tls in ((x[:h], il[:h], tg[:h], tl[:h]),
                              (torch.cat([x[:h], x[:h]]), torch.cat([il[:h], il[:h]]), torch.cat([tg[:h], tg[:h]]), torch.cat([tl[:h], tl[:h]]))):
        for p in list(enc.parameters()) + list(rec.parameters()):
            p.grad = None
        feats, flen, _ = enc(xs.contiguous(), ils)
        loss, _ = rec(feats, tgs, flen, tls)
        loss.backward()
        out.append((loss.item(), {n: p.grad.clone() for n, p in list(enc.named_parameters()) + list(rec.named_parameters())}))
    (l1, g1), (l2, g2) = out
    tol = 2e-6 if math_mode == 'bf16x3' else 2e-3          # bf16: another batch shape rounds the gate-gradient operands at other values
    assert abs(l1 - l2) <= 1e-6 * abs(l1)
    for n in g1:
        scale = g1[n].abs().max().item()
        assert (g1[n] - g2[n]).abs().max().item() <= tol * scale + 1e-12, n
    db = g2['classifier.bias']
    assert abs(db.sum().item()) <= 1e-5 * db.abs().sum().item()


def test_a_batch_of_two_launches_is_the_two_batches(model):
    """B = 128 runs the two-layer persistent launches twice over the same buffers (csrc/lstm_persist2.hip, launch_groups): its features are
    those of its two halves run as batches of 64 -- bit for bit, also with the halves swapped -- and so are the per-utterance losses."""
    lib, enc, rec, (x, il, tg, tl) = model
    prev = lib.get_math_mode()
    lib.set_math_mode('bf16')
    try:
        x2 = torch.randn(B, T, F_, generator=torch.Generator().manual_seed(11)).to(DEV)
        xx, ii = torch.cat([x, x2]), torch.cat([il, il.flip(0)])
        assert lib.lib().halo_lstm_persistent2_eligible(21, 2 * B, H, L) == 1
        with torch.no_grad():
            fa, la, _ = enc(x, il)
            fb, lb, _ = enc(x2, il.flip(0))
            fab, lab, _ = enc(xx, ii)
            fba, _, _ = enc(torch.cat([x2, x]), torch.cat([il.flip(0), il]))
        assert torch.equal(lab, torch.cat([la, lb]))
        assert torch.equal(fab, torch.cat([fa, fb])) and torch.equal(fba, torch.cat([fb, fa]))
        with torch.no_grad():
            n_ab = _nll(rec, fab, torch.cat([tg, tg]), lab, torch.cat([tl, tl]))
            n_a = _nll(rec, fa, tg, la, tl)
        # (the classifier's product is tiled by the batch it sees: same sums, another order)
        np.testing.assert_allclose(n_ab[:B].cpu().numpy(), n_a.cpu().numpy(), rtol=2e-5)
    finally:
        lib.set_math_mode(prev)


def test_beam_search_properties_at_full_size(model):
    """CTC beam search (beam 16) on the model's own log-probs, all 64 utterances: scores come out sorted (descending) for every utterance;
    the batched entry gives what one call per utterance gives, and a permuted batch the permuted result -- bit for bit; hypotheses of an
    utterance are distinct; the best hypothesis scores at least the greedy path's own probability."""
    lib, enc, rec, (x, il, tg, tl) = model
    from haloop_amd import beam
    prev = lib.get_math_mode()
    lib.set_math_mode('bf16')
    try:
        with torch.no_grad():
            feats, flen, _ = enc(x, il)
            lp = rec.log_probs(feats).contiguous()
        hyps, scores = beam.decode_batch(lp, 16, True)
        sc = scores.cpu().numpy()
        assert np.isfinite(sc).all() and (np.diff(sc, axis=1) <= 0).all()
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(2))
        hyps_p, scores_p = beam.decode_batch(lp[perm.to(DEV)].contiguous(), 16, True)
        assert torch.equal(scores[perm.to(DEV)], scores_p)
        for i, j in enumerate(perm.tolist()):
            assert hyps[j] == hyps_p[i]
        greedy = lp.max(-1).values.sum(-1).cpu().numpy()           # log-probability of the single best alignment
        for n in (0, 17, 63):
            one, s_one = beam.ctc_beam_search_decode_logits(lp[n], 16)
            assert one == hyps[n] and torch.equal(s_one, scores[n])
        for n in range(B):
            assert len({tuple(h) for h in hyps[n]}) == 16
            assert sc[n, 0] >= greedy[n] - 1e-4
    finally:
        lib.set_math_mode(prev)


@pytest.mark.parametrize('mode', ['bf16', 'bf16x3'])
def test_gpt2_small_is_causal_and_row_independent(mode):
    """BASELINE config 3 (GPT-2 small, 12 layers x 12 heads x 768, T = 1024, vocab 50304; random weights): the per-token losses of a
    sequence do not depend on the other sequences of the batch (each row scored alone gives the same numbers up to fp32 regrouping of the
    lm_head product), nor on any token after the position -- replacing the inputs from position 600 on leaves the first 600 per-token
    losses of every row bit-identical and changes later ones."""
    from haloop_amd import _lib, attention, synth
    _lib.lib(); _lib.lend_scratch()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        torch.manual_seed(0)
        cfg = attention.GPTConfig(block_size=1024, vocab_size=50304, n_layer=12, n_head=12, n_embd=768, dropout=0.0, bias=False)
        model = attention.GPT(cfg).to(DEV).eval()
        N, T, cut = 4, 1024, 600
        inputs, targets = synth.synthetic_tokens(N, T, 50304, 5, pad_tail=False)
        inputs, targets = inputs.to(DEV), targets.to(DEV)
        with torch.no_grad():
            base = model.forward_all(inputs, targets, reduction='none').view(N, T)
            changed = inputs.clone()
            changed[:, cut:] = torch.randint(1, 50304, (N, T - cut), generator=torch.Generator().manual_seed(6)).to(DEV)
            other = model.forward_all(changed, targets, reduction='none').view(N, T)
            assert torch.isfinite(base).all()
            assert torch.equal(base[:, :cut], other[:, :cut])
            assert not torch.equal(base[:, cut:], other[:, cut:])
            alone = model.forward_all(inputs[2:3].contiguous(), targets[2:3].contiguous(), reduction='none').view(1, T)
        np.testing.assert_allclose(alone[0].cpu().numpy(), base[2].cpu().numpy(), rtol=3e-5 if mode == 'bf16x3' else 2e-3, atol=1e-5 if mode == 'bf16x3' else 2e-3)
    finally:
        _lib.set_math_mode(prev)
