"""Pin the CPU oracle (oracle/) against fixtures produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cpu_ref, lattice, philox
from conftest import load_golden


def _params(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(prefix)}


def _unpad(seqs, lens):
    return [list(map(int, s[:n])) for s, n in zip(seqs, lens)]


@pytest.mark.parametrize('name', ['g1_tiny_l2', 'g1_tiny_l3'])
def test_model_forward_backward_matches_reference(name):
    g = load_golden(name)
    enc = {k: v.requires_grad_(True) for k, v in _params(g, 'encoder.').items()}
    rec = {k: v.requires_grad_(True) for k, v in _params(g, 'recognizer.').items()}
    x, il, tg, tl = (torch.from_numpy(g[k]) for k in ('x', 'il', 'tg', 'tl'))
    loss, feats, flen = cpu_ref.lstm_ctc_loss(enc, rec, x, il, tg, tl)
    assert flen.dtype == torch.int32
    assert np.array_equal(flen.numpy(), g['flen'])
    np.testing.assert_allclose(feats.detach().numpy(), g['feats'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    loss.backward()
    for k, v in list(enc.items()) + list(rec.items()):
        key = 'grad.' + ('encoder.' if k in enc else 'recognizer.') + k
        np.testing.assert_allclose(v.grad.numpy(), g[key], rtol=1e-4, atol=1e-7, err_msg=key)
    lp = cpu_ref.classifier_log_probs(rec, feats.detach())
    np.testing.assert_allclose(lp.detach().numpy(), g['lp'], atol=1e-6)
    hyps, hlen, ali, scores = lattice.greedy_decode(lp.detach())
    assert np.array_equal(ali.numpy(), g['ali'])
    assert np.array_equal(hlen.numpy(), g['hlen'])
    assert hyps == _unpad(g['hyps'], g['hlen'])


def test_lc2x1024_matches_reference():
    g = load_golden('g1_lc2x1024')
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], c['seed'])
    x, _, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], c['seed'])
    il = torch.from_numpy(g['il'])
    with torch.no_grad():
        loss, feats, flen = cpu_ref.lstm_ctc_loss(enc, rec, x, il, tg, tl)
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(feats[:, :, ::61].numpy(), g['feats_slice'], atol=1e-6)
    assert np.array_equal(flen.numpy(), g['flen'])


def test_lc2x1024_b64_matches_reference():
    """BASELINE config 2's shape (B=64, ragged lengths): oracle == reference on loss, feature slices and gradient norms/slices."""
    g = load_golden('g1_lc2x1024_b64')
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], c['seed'])
    x, _, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], c['seed'])
    il = torch.from_numpy(g['il'])
    pe = {k: v.clone().requires_grad_(True) for k, v in enc.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in rec.items()}
    loss, feats, flen = cpu_ref.lstm_ctc_loss(pe, pr, x, il, tg, tl)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(feats.detach()[:, :, ::61].numpy(), g['feats_slice'], atol=1e-6)
    assert np.array_equal(flen.numpy(), g['flen'])
    for prefix, d in (('encoder.', pe), ('recognizer.', pr)):
        for k, v in d.items():
            np.testing.assert_allclose(v.grad.double().norm().item(), float(g['gradnorm.' + prefix + k]), rtol=1e-5, err_msg=k)
            np.testing.assert_allclose(v.grad.reshape(-1)[::9973].numpy(), g['gradslice.' + prefix + k], rtol=1e-4, atol=1e-8, err_msg=k)


def test_stock3_b64_matches_reference():
    """The reference's stock 3-layer encoder (ha/rnn.py:6-11, nothing swapped) at B=64 with ragged lengths: oracle == reference on loss,
    feature slices, greedy alignments and every gradient's norm and slice."""
    g = load_golden('g1_stock3_b64')
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    assert c['L'] == 3
    enc, rec = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], c['seed'])
    x, _, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], c['seed'])
    il = torch.from_numpy(g['il'])
    pe = {k: v.clone().requires_grad_(True) for k, v in enc.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in rec.items()}
    loss, feats, flen = cpu_ref.lstm_ctc_loss(pe, pr, x, il, tg, tl)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(feats.detach()[:, :, ::61].numpy(), g['feats_slice'], atol=1e-6)
    assert np.array_equal(flen.numpy(), g['flen'])
    assert set('encoder.' + k for k in pe) | set('recognizer.' + k for k in pr) == set(k[9:] for k in g if k.startswith('gradnorm.'))
    for prefix, d in (('encoder.', pe), ('recognizer.', pr)):
        for k, v in d.items():
            np.testing.assert_allclose(v.grad.double().norm().item(), float(g['gradnorm.' + prefix + k]), rtol=1e-5, err_msg=k)
            np.testing.assert_allclose(v.grad.reshape(-1)[::9973].numpy(), g['gradslice.' + prefix + k], rtol=1e-4, atol=1e-8, err_msg=k)


def test_train_steps_b64_match_reference():
    g = load_golden('g1_train3_b64')
    c = {k[4:]: v for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = cpu_ref.make_params(int(c['F_']), int(c['C']), int(c['H']), int(c['L']), int(c['V']), int(c['seed']))
    tr = cpu_ref.Trainer(enc, rec, lr=float(c['lr']))
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(int(c['B']), int(c['T']), int(c['F_']), int(c['V']), int(c['S']), 200 + step)
        loss, gn = tr.step(x, il, tg, tl)
        np.testing.assert_allclose(float(loss), g['losses'][step], rtol=1e-5)
        np.testing.assert_allclose(float(gn), g['gnorms'][step], rtol=1e-4)
    for prefix, d in (('encoder.', tr.enc), ('recognizer.', tr.rec)):
        for k, v in d.items():
            np.testing.assert_allclose(v.detach().reshape(-1)[::4999].numpy(), g['finalslice.' + prefix + k], atol=2e-6, err_msg=k)


def test_train_steps_match_reference():
    g = load_golden('g1_train3')
    c = {k[4:]: v for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = cpu_ref.make_params(int(c['F_']), int(c['C']), int(c['H']), int(c['L']), int(c['V']), int(c['seed']))
    tr = cpu_ref.Trainer(enc, rec, lr=float(c['lr']))
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(int(c['B']), int(c['T']), int(c['F_']), int(c['V']), int(c['S']), 100 + step)
        loss, gn = tr.step(x, il, tg, tl)
        np.testing.assert_allclose(float(loss), g['losses'][step], rtol=1e-5)
        np.testing.assert_allclose(float(gn), g['gnorms'][step], rtol=1e-4)
    for k, v in tr.enc.items():
        np.testing.assert_allclose(v.detach().numpy(), g['final.encoder.' + k], atol=2e-6, err_msg=k)
    for k, v in tr.rec.items():
        np.testing.assert_allclose(v.detach().numpy(), g['final.recognizer.' + k], atol=2e-6, err_msg=k)


@pytest.mark.parametrize('case', ['random', 'repeat', 's1', 'ragged', 'infeasible', 'wide'])
def test_ctc_score3_matches_reference(case):
    g = load_golden('g2_ctc')
    em = torch.from_numpy(g[case + '.logits']).log_softmax(-1)
    tg, il, tl = (torch.from_numpy(g[case + '.' + k]) for k in ('targets', 'il', 'tl'))
    nll = lattice.ctc_forward_score3(em, tg, il, tl)
    np.testing.assert_array_equal(nll.numpy(), g[case + '.score3'])          # same ATen ops -> bit equal
    np.testing.assert_array_equal(lattice.ctc_reduce_mean(nll, tl).numpy(), g[case + '.reduce_mean3'])
    ok = np.isfinite(g[case + '.torch_none'])
    np.testing.assert_allclose(nll.numpy()[ok], g[case + '.torch_none'][ok], rtol=2e-5)
    if case == 'infeasible':
        assert g['infeasible.torch_none'][0] == np.inf and nll[0] == torch.finfo(torch.float32).max


def test_ctc_single_sequence_variants():
    g = load_golden('g2_ctc')
    l0, t0 = torch.from_numpy(g['demo.l0']), torch.from_numpy(g['demo.t0'])
    np.testing.assert_allclose(float(lattice.ctc_forward_score1(l0, t0)), float(g['demo.score1']), rtol=1e-6)
    np.testing.assert_allclose(float(lattice.ctc_forward_score2(l0, t0)), float(g['demo.score2']), rtol=1e-6)
    assert abs(float(g['demo.score1']) - 7.6325) < 1e-4          # ha/ctc.py __main__ print, seed 2
    em, tg = torch.from_numpy(g['demo.em']), torch.from_numpy(g['demo.tg'])
    s3 = lattice.ctc_forward_score3(em, tg, torch.tensor([5, 5]), torch.tensor([3, 4]))
    np.testing.assert_array_equal(s3.numpy(), g['demo.score3'])
    lw, tw = torch.from_numpy(g['wrap.l']), torch.from_numpy(g['wrap.t'])
    np.testing.assert_allclose(float(lattice.ctc_forward_score1(lw, tw)), float(g['wrap.score1']), rtol=1e-6)
    np.testing.assert_allclose(float(lattice.ctc_forward_score2(lw, tw)), float(g['wrap.score2']), rtol=1e-6)
    assert abs(float(g['wrap.score1']) - float(g['wrap.score2'])) > 1.0   # the wrap quirk is real


def test_ctc_lattice_equals_brute_force():
    gen = torch.Generator().manual_seed(3)
    lp = torch.randn(5, 3, generator=gen).log_softmax(-1)
    for target in ([1], [1, 2], [2, 2], [1, 2, 1]):
        tg = torch.tensor([target])
        nll = lattice.ctc_forward_score3(lp[:, None, :], tg, torch.tensor([5]), torch.tensor([len(target)]))
        assert abs(float(nll[0]) - lattice.ctc_brute_force_nll(lp, target)) < 1e-5


@pytest.mark.parametrize('case', ['onehot', 'r21x32b16', 'r21x32b3', 'r6x4b4', 'r30x9b5', 'r21x32b33',
                                  'p21x32b16', 'p40x32b8', 'p21x256b4', 'p64x9b9'])
def test_beam_logits_matches_reference(case):
    g = load_golden('g3_beam')
    seqs, scores = lattice.ctc_beam_search_decode_logits(torch.from_numpy(g[case + '.logits']), int(g[case + '.beam']))
    assert seqs == _unpad(g[case + '.seqs'], g[case + '.lens'])
    np.testing.assert_array_equal(scores.numpy(), g[case + '.scores'])


def test_beam_probs_matches_patched_reference():
    g = load_golden('g3_beam')
    assert bool(g['probs.raises_nameerror'])        # as shipped, ha/beam.py:46 raises
    seqs, scores = lattice.ctc_beam_search_decode_probs(torch.from_numpy(g['probs.probs']), int(g['probs.beam']))
    assert seqs == _unpad(g['probs.seqs'], g['probs.lens'])
    np.testing.assert_allclose(scores.numpy(), g['probs.scores'], rtol=1e-6)
    with pytest.raises(RuntimeError):
        lattice.ctc_beam_search_decode_logits(torch.zeros(3, 4).log_softmax(-1), beam_size=6)
    assert bool(g['toowide.raises'])


def test_philox_known_answers():
    # Random123 kat_vectors: philox4x32-10, ctr=0 key=0 and ctr=key=ffffffff
    z = np.zeros(1, dtype=np.uint32)
    r = philox.philox4x32_10(z, z, z, z, 0, 0)
    assert [int(v[0]) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    f = np.full(1, 0xffffffff, dtype=np.uint32)
    r = philox.philox4x32_10(f, f, f, f, 0xffffffff, 0xffffffff)
    assert [int(v[0]) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    m = philox.dropout_mask(100000, 0.2, 1234, 1, 0)
    assert abs((m == 0).mean() - 0.2) < 0.01
    assert set(np.unique(m)) == {np.float32(0), np.float32(1.25)}


@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias', 'g5_gpt_tiny_stable', 'g5_gpt_tiny_bidir', 'g5_gpt2_small'])
def test_gpt_forward_all_matches_reference(name):
    from oracle import gpt_ref
    g = load_golden(name)
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    if name == 'g5_gpt2_small':
        params = gpt_ref.make_gpt_params(vocab, block, n_layer, n_head, n_embd, bool(bias), seed)
    else:
        params = {k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}
    inputs, targets = torch.from_numpy(g['inputs']), torch.from_numpy(g['targets'])
    with torch.no_grad():
        causal = bool(g.get('causal', 1))
        per_tok = gpt_ref.gpt_forward_all(params, n_layer, n_head, inputs, targets, reduction='none', causal=causal)
        mean = gpt_ref.gpt_forward_all(params, n_layer, n_head, inputs, targets, reduction='mean', causal=causal)
    np.testing.assert_allclose(per_tok.numpy(), g['per_token'], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(mean), float(g['mean']), rtol=1e-6)
    i2, t2 = gpt_ref.synthetic_tokens(B, T, vocab, seed + 1)
    assert torch.equal(i2, inputs) and torch.equal(t2, targets)


@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias'])
def test_gpt_kv_cache_forward_matches_reference(name):
    from oracle import gpt_ref
    g = load_golden(name)
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    params = {k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}
    inputs, split = torch.from_numpy(g['inputs']), int(g['gen.split'])
    with torch.no_grad():
        l0, past = gpt_ref.gpt_forward(params, n_layer, n_head, inputs[:, :split])
        l1, past = gpt_ref.gpt_forward(params, n_layer, n_head, inputs[:, split:split + 1], past)
        l2, past = gpt_ref.gpt_forward(params, n_layer, n_head, inputs[:, split + 1:split + 4], past)
    for got, key in ((l0, 'gen.logits0'), (l1, 'gen.logits1'), (l2, 'gen.logits2'), (past, 'gen.present2')):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=2e-6, err_msg=key)


@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias', 'g5_gpt_tiny_stable', 'g5_gpt_tiny_bidir'])
def test_gpt_gradients_match_reference(name):
    """The training direction: autograd through the restatement == the reference's loss.backward()."""
    from oracle import gpt_ref
    g = load_golden(name)
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    params = {k[len('param.'):]: torch.from_numpy(v).requires_grad_(True) for k, v in g.items() if k.startswith('param.')}
    params['lm_head.weight'] = params['transformer.wte.weight']
    inputs, targets = torch.from_numpy(g['inputs']), torch.from_numpy(g['targets'])
    gpt_ref.gpt_forward_all(params, n_layer, n_head, inputs, targets, reduction='mean', causal=bool(g.get('causal', 1))).backward()
    checked = 0
    for k, v in g.items():
        if k.startswith('grad.'):
            np.testing.assert_allclose(params[k[len('grad.'):]].grad.numpy(), v, rtol=1e-4, atol=1e-7, err_msg=k)
            checked += 1
    assert checked >= 10


# ------------------------------------------------------------------------------ enc-dec attention ASR (ha/transformer.py)
ASR_CASES = ['g6_asr_tiny', 'g6_asr_tiny_s221', 'g6_asr_tiny_stop', 'g6_asr_transformer32', 'g6_asr_transformer32_stop']


def asr_case_from_golden(name):
    from oracle import transformer_ref as tr
    g = load_golden(name)
    vocab, hd, heads, el, dl, conv_dim, N, T, S, seed, F_ = (int(v) for v in g['cfg'])
    strides = tuple(int(s) for s in g['strides'])
    pe = tr.make_encoder_params(hd, heads, el, F_, conv_dim, len(strides), seed)
    pd = tr.make_decoder_params(vocab, hd, heads, dl, seed + 1)
    x, il, tg, tl = tr.synthetic_asr_batch(N, T, F_, vocab, S, seed + 2)
    return g, pe, pd, (x, il, tg, tl), heads, strides


@pytest.mark.parametrize('name', ASR_CASES)
def test_asr_encoder_decoder_matches_reference(name):
    from oracle import transformer_ref as tr
    g, pe, pd, (x, il, tg, tl), heads, strides = asr_case_from_golden(name)
    N = x.shape[0]
    with torch.no_grad():
        feats, flen = tr.audio_encoder_forward(pe, x, il, heads, strides)
        assert flen.dtype == torch.int32 and np.array_equal(flen.numpy(), g['feature_lengths'])
        np.testing.assert_allclose(feats.numpy(), g['features'], rtol=0, atol=2e-5)
        feats = torch.from_numpy(g['features'])
        for red in ('mean', 'none', 'sumeach'):
            loss = tr.decoder_forward(pd, feats, tg, flen, tl, heads, reduction=red, pre='decoder.')
            np.testing.assert_allclose(loss.numpy(), g['decoder_loss.' + red], rtol=2e-5, atol=2e-5, err_msg=red)
        cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1)
        joint, _, _ = tr.ctc_attention_forward(pd, feats, cond, flen, tl + 1, heads)
        np.testing.assert_allclose(float(joint), float(g['joint_loss']), rtol=1e-5)
        ents = []
        tr.decoder_forward(pd, feats, tg, flen, tl, heads, pre='decoder.', entropies=ents)
        np.testing.assert_allclose(np.array([float(m) for m, _ in ents]), g['meme_entropy'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(np.array([float(t) for _, t in ents]), g['self_entropy'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('name', ASR_CASES)
def test_asr_greedy_decode_matches_reference_fp16_autocast(name):
    """The reference's decode only runs under fp16 autocast; the oracle keeps fp16-rounded caches and fp32
    arithmetic: tokens and lengths exact on these decisive fixtures, accumulated scores to fp16 tolerance."""
    from oracle import transformer_ref as tr
    g, pe, pd, (x, il, tg, tl), heads, strides = asr_case_from_golden(name)
    feats, flen = torch.from_numpy(g['features']), torch.from_numpy(g['feature_lengths'])
    with torch.no_grad():
        outs, olen, lps, ents, _ = tr.decoder_decode(pd, feats, flen, tl, heads, pre='decoder.')
        assert olen.dtype == torch.int32 and np.array_equal(olen.numpy(), g['decode.output_lengths'])
        assert [o.tolist() for o in outs] == _unpad(g['decode.tokens'], g['decode.token_lens'])
        np.testing.assert_allclose(lps.numpy(), g['decode.log_probs'], rtol=1e-2, atol=5e-2)
        np.testing.assert_allclose(ents.numpy(), g['decode.sum_entropies'], rtol=1e-2, atol=5e-2)
        outs, olen, lps, _, _ = tr.decoder_decode(pd, feats, flen, tl, heads, prompt=torch.tensor([[7, 9]] * len(tl)), pre='decoder.')
        assert np.array_equal(olen.numpy(), g['decode_prompt.output_lengths'])
        assert [o.tolist() for o in outs] == _unpad(g['decode_prompt.tokens'], g['decode_prompt.token_lens'])
        np.testing.assert_allclose(lps.numpy(), g['decode_prompt.log_probs'], rtol=1e-2, atol=5e-2)


def test_asr_parts_match_reference():
    from oracle import transformer_ref as tr
    g = load_golden('g6_asr_parts')
    for nm in 'abc':
        y = tr.rotate_interleaved(torch.from_numpy(g[f'rope.{nm}.x']), t0=int(g[f'rope.{nm}.t0']))
        np.testing.assert_allclose(y.numpy(), g[f'rope.{nm}.y'], rtol=0, atol=1e-6)
    y, ent = tr.attend(*(torch.from_numpy(g['attend.' + k]) for k in ('q', 'k', 'v', 'mask')))
    np.testing.assert_allclose(y.numpy(), g['attend.y'], atol=1e-6)
    np.testing.assert_allclose(float(ent), float(g['attend.entropy']), rtol=1e-6)
    lens = torch.from_numpy(g['lengths.in'])
    assert np.array_equal(tr.subsampled_lengths(lens, (2, 2, 2)).numpy(), g['lengths.s222'])
    assert np.array_equal(tr.subsampled_lengths(lens, (2, 2, 1)).numpy(), g['lengths.s221'])


@pytest.mark.parametrize('name', ['g6_asr_tiny', 'g6_asr_tiny_s221'])
def test_asr_gradients_match_reference(name):
    """Training direction: autograd through the restatement (encoder -> decoder CE + 0.3 CTC) == the reference's backward."""
    from oracle import transformer_ref as tr
    g, pe, pd, (x, il, tg, tl), heads, strides = asr_case_from_golden(name)
    pe = {k: v.requires_grad_(True) for k, v in pe.items()}
    pd = {k: v.requires_grad_(True) for k, v in pd.items()}
    feats, flen = tr.audio_encoder_forward(pe, x, il, heads, strides)
    cond = torch.cat([torch.full((x.shape[0], 1), 5, dtype=torch.long), tg], dim=1)
    joint, _, _ = tr.ctc_attention_forward(pd, feats, cond, flen, tl + 1, heads)
    np.testing.assert_allclose(float(joint), float(g['train.joint_loss']), rtol=1e-5)
    joint.backward()
    n = 0
    for pre, params in (('encoder.', pe), ('decoder.', pd)):
        for k, v in params.items():
            want = g['grad.' + pre + k]
            np.testing.assert_allclose(v.grad.numpy(), want, rtol=2e-3, atol=1e-5 * max(1.0, float(np.abs(want).max())), err_msg=k)
            n += 1
    assert n == sum(1 for k in g if k.startswith('grad.'))


@pytest.mark.parametrize('name', ['g7_audio_encoder_tiny', 'g7_audio_encoder_bias'])
def test_audio_encoder_matches_reference(name):
    """ha.attention_audio.AudioEncoder (rotary_emb_dim = 0) + CTC head: oracle == reference on features, lengths, loss, gradients."""
    from oracle import audio_encoder_ref as ae
    g = load_golden(name)
    d_input, n_embd, n_head, n_layer, block, bias, vocab, B, T, S, seed = (int(v) for v in g['cfg'])
    p = {k: v.clone().requires_grad_(k != 'transformer.wpe.weight') for k, v in ae.make_params(d_input, n_embd, n_layer, block, bool(bias), seed).items()}
    np.testing.assert_array_equal(p['transformer.wpe.weight'][:8].detach().numpy(), g['wpe_head'])
    rec_p, x, il, tg, tl = ae.make_head_and_batch(n_embd, vocab, d_input, B, T, S, seed)
    rec_p = {k: v.clone().requires_grad_(True) for k, v in rec_p.items()}
    feats, flen, _ = ae.forward(p, n_layer, n_head, x, il)
    feats.retain_grad()
    loss, _ = cpu_ref.classifier_loss(rec_p, feats, tg, flen, tl)
    loss.backward()
    assert flen.dtype == torch.int32 and np.array_equal(flen.numpy(), g['flen'])
    np.testing.assert_allclose(feats.detach().numpy(), g['feats'], atol=2e-5)
    np.testing.assert_allclose(float(loss), float(g['loss']), rtol=1e-6)
    np.testing.assert_allclose(feats.grad.numpy(), g['dfeats'], rtol=1e-4, atol=1e-6)
    for k, v in list(p.items()) + [('rec:' + k, v) for k, v in rec_p.items()]:
        if v.grad is None:
            continue
        key = ('recgrad.' + k[4:]) if k.startswith('rec:') else 'grad.' + k
        if key in g:
            np.testing.assert_allclose(v.grad.numpy(), g[key], rtol=1e-3, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_allclose(float(v.grad.double().norm()), float(g['norm.' + key]), rtol=1e-4, err_msg=k)
            np.testing.assert_allclose(v.grad.reshape(-1)[::97].numpy(), g['slice.' + key], rtol=1e-3, atol=1e-6, err_msg=k)


def test_symbol_tape_restatement_on_the_reference_demo():
    """ha/symbol_tape.py:311-313 (`__main__`): the 48-letter tape, batch_size 2, bptt_len 8.  tape_len = 24, three parts, no
    trailing part; column b of part i is data[b*23 + 8i : b*23 + 8i + 8] (worked out by hand from :252-277)."""
    from oracle import tape_ref
    data = np.frombuffer(b'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuv', dtype=np.uint8)
    tape = tape_ref.SymbolTapeNoPad(data, batch_size=2, bptt_len=8)
    assert (tape.tape_len, tape.tape_parts, tape.trailing_tokens, len(tape)) == (24, 3, 0, 3)
    assert tape[0].shape == (8, 2)
    assert tape[0].T.tobytes() == b'ABCDEFGHXYZabcde'
    assert tape[1].T.tobytes() == b'IJKLMNOPfghijklm'
    assert tape[2].T.tobytes() == b'QRSTUVWXnopqrstu'
    # a ragged tape: 50 tokens, 3 columns -> tape_len 17, parts of 5 with 2 trailing rows; the last column runs off the end
    tape = tape_ref.SymbolTapeNoPad(np.arange(1, 51, dtype=np.int16), batch_size=3, bptt_len=5)
    assert (tape.tape_len, tape.tape_parts, tape.trailing_tokens, len(tape)) == (17, 3, 2, 4)
    assert tape[3].tolist() == [[16, 32, 48], [17, 33, 49]]
    assert tape[2][:, 2].tolist() == [43, 44, 45, 46, 47]
    x, y = tape_ref.get_batch(np.arange(100, 200, dtype=np.uint16), [0, 90], 8)
    assert x[1].tolist() == list(range(190, 198)) and y[1].tolist() == list(range(191, 198)) + [0]


def test_product_side_synthetic_generators_match_the_oracle_side():
    """haloop_amd/synth.py (what bench.py's measured legs use) and the oracle's generators (what the parity tests use) must draw
    the same tensors from the same seeds."""
    from haloop_amd import synth
    from oracle import cpu_ref, gpt_ref
    for (a, b) in zip(synth.make_params(12, 16, 32, 2, 9, 5), cpu_ref.make_params(12, 16, 32, 2, 9, 5)):
        assert list(a) == list(b)
        for k in a:
            assert torch.equal(a[k], b[k]), k
    for a, b in zip(synth.synthetic_batch(4, 41, 12, 9, 4, 7), cpu_ref.synthetic_batch(4, 41, 12, 9, 4, 7)):
        assert torch.equal(a, b)
    for pad in (True, False):
        for a, b in zip(synth.synthetic_tokens(3, 24, 61, 9, pad_tail=pad), gpt_ref.synthetic_tokens(3, 24, 61, 9, pad_tail=pad)):
            assert torch.equal(a, b)


# ---- star-CTC and transducer lattices (ha/star.py, ha/transducer.py): oracle/star_ref.py against the reference's own outputs ----
STAR_CASES = ['random', 'repeat', 's1', 'nopenalty', 'padded']
TRANSDUCER_CASES = ['batched', 'ragged', 'long']


def star_case_from_golden(g, name):
    t = lambda k: torch.from_numpy(g[name + '.' + k])
    return t('emissions'), t('targets'), t('il'), t('tl'), float(g[name + '.penalty'])


@pytest.mark.parametrize('name', STAR_CASES + ['demo'])
def test_star_ctc_restatement_matches_reference(name):
    from oracle import star_ref
    g = load_golden('g8_star')
    em, tg, il, tl, pen = star_case_from_golden(g, name)
    losses = star_ref.star_ctc_forward_score(em, tg, il, tl, star_penalty=pen)
    np.testing.assert_allclose(losses.numpy(), g[name + '.losses'], rtol=2e-6, atol=1e-5)
    if name == 'demo':
        return
    C = em.shape[-1]
    assert np.array_equal(star_ref.star_states(tg, C)[:, 1::2].numpy(), g[name + '.star_targets'])       # the non-blank states
    np.testing.assert_allclose(star_ref.star_emissions(em)[0].numpy(), g[name + '.star_emissions_t0'], rtol=1e-6, atol=1e-6)
    # the analytic alpha-beta gradient against the reference's autograd gradient of sum(losses)
    grad = star_ref.star_ctc_grad(em, tg, il, tl, star_penalty=pen)
    np.testing.assert_allclose(grad.numpy(), g[name + '.grad'], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize('name', TRANSDUCER_CASES)
def test_transducer_restatement_matches_reference(name):
    from oracle import star_ref
    g = load_golden('g9_transducer')
    t = lambda k: torch.from_numpy(g[name + '.' + k])
    joint, tg, jl, tl = t('joint'), t('targets'), t('jl'), t('tl')
    if name == 'long':                                      # pure-Python cells: keep the big lattice to its first sequence
        joint, tg, jl, tl = joint[:1], tg[:1], jl[:1], tl[:1]
    n = joint.shape[0]
    losses = star_ref.transducer_forward_score(joint, tg, jl, tl)
    np.testing.assert_allclose(losses.numpy(), g[name + '.losses'][:n], rtol=1e-5, atol=1e-5)
    grad = star_ref.transducer_grad(joint, tg, jl, tl)
    np.testing.assert_allclose(grad.numpy(), g[name + '.grad'][:n], rtol=1e-4, atol=2e-6)
    if name == 'batched':
        np.testing.assert_allclose(losses[0].numpy(), g['batched.score4_seq0'], rtol=1e-5)


def test_fbank_restatement_matches_the_transformers_port():
    """oracle/fbank_ref.py against g11_fbank: vectors of the Hugging Face transformers port of torchaudio.compliance.kaldi.fbank (the
    numpy path of Speech2TextFeatureExtractor; tests/golden/make_golden.py save_fbank).  torchaudio itself, the reference's dependency,
    is not installable here, so this pins the restatement to an independent published implementation of the same function rather than
    to torchaudio's own output.  float32 log-mels of magnitude ~25: 4e-6 abs = 2 ulp."""
    from oracle import fbank_ref
    g = load_golden('g11_fbank')
    for k in ('tone', 'noise', 'chirp', 'one_frame'):
        got = fbank_ref.fbank(g[f'{k}.wav'], num_mel_bins=80)
        assert got.shape == g[f'{k}.fbank'].shape
        np.testing.assert_allclose(got, g[f'{k}.fbank'], rtol=0, atol=4e-6, err_msg=k)


def test_fbank_restatement_properties():
    """What can be checked of oracle/fbank_ref.py without any vectors: frame count, the filter bank's
    shape / support / partition of unity between the first and last centre, a pure tone landing in the filter that covers it, and
    invariance to a DC offset."""
    from oracle import fbank_ref
    banks = fbank_ref.mel_banks(80, 512, 16000.0)
    assert banks.shape == (80, 257) and (banks >= 0).all() and (banks[:, 256] == 0).all()
    mel = fbank_ref.mel_scale(31.25 * np.arange(256))
    lo, hi = fbank_ref.mel_scale(20.0), fbank_ref.mel_scale(8000.0)
    delta = (hi - lo) / 81
    inside = (mel >= lo + delta) & (mel <= hi - delta)
    np.testing.assert_allclose(banks[:, :256].sum(0)[inside], 1.0, atol=1e-9)      # neighbouring triangles sum to one
    t = np.arange(16000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    f = fbank_ref.fbank(tone, num_mel_bins=80)
    assert f.shape == (98, 80)
    centres = 700.0 * (np.exp((lo + (np.arange(80) + 1) * delta) / 1127.0) - 1.0)
    assert abs(centres[int(f.mean(0).argmax())] - 1000.0) < 60.0
    np.testing.assert_allclose(fbank_ref.fbank(tone + 0.25, num_mel_bins=80), f, atol=2e-3)
    assert fbank_ref.fbank(tone[:399], num_mel_bins=80).shape == (0, 80)
