"""GPU parity of haloop_amd.attention_audio.AudioEncoder (ha/attention_audio.py:64-117, rotary_emb_dim = 0: the `audio-encoder` arch
that BASELINE config 5 names) against reference-generated fixtures (g7_*) and, in training mode, against the CPU oracle fed the
same Philox masks."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture
def math_mode(request):
    from haloop_amd import _lib
    _lib.lib(); _lib.lend_scratch()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(request.param)
    yield request.param
    _lib.set_math_mode(prev)


BOTH_MODES = pytest.mark.parametrize('math_mode', ['f32', 'bf16x3'], indirect=True)


def _build(g, dropout=0.0):
    from haloop_amd import attention, attention_audio, recognizer
    from oracle import audio_encoder_ref as ae
    d_input, n_embd, n_head, n_layer, block, bias, vocab, B, T, S, seed = (int(v) for v in g['cfg'])
    cfg = attention.GPTConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head, n_embd=n_embd, bias=bool(bias),
                              causal=False, d_input=d_input, rotary_emb_dim=0, dropout=dropout)
    enc = attention_audio.AudioEncoder(cfg)
    params = ae.make_params(d_input, n_embd, n_layer, block, bool(bias), seed)
    enc.load_state_dict(params, strict=True)
    rec_p, x, il, tg, tl = ae.make_head_and_batch(n_embd, vocab, d_input, B, T, S, seed)
    rec = recognizer.TemporalClassifier(n_embd, vocab)
    rec.load_state_dict(rec_p)
    return enc.to(DEV), rec.to(DEV), params, rec_p, (x, il, tg, tl), (n_layer, n_head, n_embd, B)


@BOTH_MODES
@pytest.mark.parametrize('name', ['g7_audio_encoder_tiny', 'g7_audio_encoder_bias'])
def test_audio_encoder_matches_reference(name, math_mode):
    g = load_golden(name)
    enc, rec, params, rec_p, (x, il, tg, tl), _ = _build(g)
    enc.eval(); rec.eval()
    assert not enc.transformer.wpe.weight.requires_grad
    with torch.no_grad():                                                     # inference path
        f0, l0, stats = enc(x.to(DEV), il.to(DEV))
    assert stats == {} and l0.dtype == torch.int32 and np.array_equal(l0.cpu().numpy(), g['flen'])
    np.testing.assert_allclose(f0.cpu().numpy(), g['feats'], atol=1e-4)
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))                               # autograd path
    feats.retain_grad()
    np.testing.assert_allclose(feats.detach().cpu().numpy(), g['feats'], atol=1e-4)
    loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=1e-5)
    loss.backward()
    np.testing.assert_allclose(feats.grad.cpu().numpy(), g['dfeats'], rtol=1e-3, atol=1e-6)
    named = [('grad.' + k, p) for k, p in enc.named_parameters() if p.requires_grad] + [('recgrad.' + k, p) for k, p in rec.named_parameters()]
    for key, p in named:
        got = p.grad.cpu().numpy()
        if key in g:
            np.testing.assert_allclose(got, g[key], rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(g[key]).max(), err_msg=key)
        else:
            np.testing.assert_allclose(float(np.sqrt((got.astype(np.float64) ** 2).sum())), float(g['norm.' + key]), rtol=2e-4, err_msg=key)
            ref = g['slice.' + key]
            np.testing.assert_allclose(got.reshape(-1)[::97], ref, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(ref).max(), err_msg=key)


@BOTH_MODES
def test_audio_encoder_training_mode_dropout_matches_oracle_with_same_masks(math_mode):
    """config.dropout = 0.1, .train(): the Philox masks of every site (embedding dropout, then attention / c_proj / MLP per block) are
    restated on the CPU and fed to the oracle."""
    from oracle import audio_encoder_ref as ae, cpu_ref, philox
    g = load_golden('g7_audio_encoder_tiny')
    P, SEED = 0.1, 0xFEEDFACE54321
    enc, rec, params, rec_p, (x, il, tg, tl), (n_layer, n_head, C, B) = _build(g, dropout=P)
    enc.train(); rec.eval()
    enc.dropout_stream.seed = SEED
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))
    loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
    loss.backward()
    T = feats.shape[1]
    rows = lambda sid: torch.from_numpy(philox.dropout_mask(B * T * C, P, SEED, sid, 0)).view(B, T, C)
    masks = {'emb': rows(64), 'att': [], 'res': [], 'mlp': []}
    for i in range(n_layer):
        masks['att'].append(torch.from_numpy(philox.attention_dropout_mask(B, n_head, T, T, P, SEED, 65 + 3 * i, 0).copy()))
        masks['res'].append(rows(66 + 3 * i))
        masks['mlp'].append(rows(67 + 3 * i))
    pr = {k: v.clone().requires_grad_(k != 'transformer.wpe.weight') for k, v in params.items()}
    rr = {k: v.clone().requires_grad_(True) for k, v in rec_p.items()}
    f_ref, flen_ref, _ = ae.forward(pr, n_layer, n_head, x, il, masks=masks)
    loss_ref, _ = cpu_ref.classifier_loss(rr, f_ref, tg, flen_ref, tl)
    loss_ref.backward()
    np.testing.assert_allclose(feats.detach().cpu().numpy(), f_ref.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-5)
    for k, p in enc.named_parameters():
        if p.requires_grad:
            ref = pr[k].grad.numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)


def test_audio_encoder_refuses_what_is_not_built():
    from haloop_amd import attention, attention_audio
    cfg = attention.GPTConfig(block_size=64, vocab_size=11, n_layer=1, n_head=2, n_embd=64, causal=False, d_input=20, rotary_emb_dim=64)
    with pytest.raises(NotImplementedError):
        attention_audio.AudioEncoder(cfg)
    with pytest.raises(NotImplementedError):
        attention_audio.StridingAudioEncoder(cfg)
    cfg.rotary_emb_dim = 0
    enc = attention_audio.AudioEncoder(cfg)
    with pytest.raises(Exception):
        enc(torch.zeros(1, 16, 20), torch.tensor([16]))             # CPU tensors: no CPU path
