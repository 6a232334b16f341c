"""GPU parity of the encoder-decoder attention ASR path (haloop_amd.transformer / haloop_amd.conv, through the
C ABI) against the reference-generated goldens (tests/golden/g6_*.npz) and the CPU oracle
(oracle/transformer_ref.py).  Tolerances (fp32 state; both GEMM modes):

    encoder features <= 2e-4 abs (12 layers; 2e-5 on the tiny nets), lengths exact (int32);
    teacher-forced losses rel <= 2e-5 (per-token 1e-4 abs); attention-entropy monitors rel 1e-4;
    greedy decode: token ids and lengths EXACT, accumulated log-probs / entropies <= 2e-3 abs against the oracle
    (same fp16-cache arithmetic) and <= 5e-2 against the reference's own fp16-autocast run.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from test_oracle_golden import ASR_CASES, asr_case_from_golden, _unpad

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def hal():
    from haloop_amd import _lib, ops, transformer, conv
    _lib.lib()
    _lib.lend_scratch()
    return dict(ops=ops, tr=transformer, conv=conv, lib=_lib)


@pytest.fixture
def math_mode(request, hal):
    prev = hal['lib'].get_math_mode()
    hal['lib'].set_math_mode(request.param)
    yield request.param
    hal['lib'].set_math_mode(prev)


BOTH_MODES = pytest.mark.parametrize('math_mode', ['f32', 'bf16x3'], indirect=True)


def _models(hal, name):
    g, pe, pd, batch, heads, strides = asr_case_from_golden(name)
    vocab, hd, heads, el, dl, conv_dim, N, T, S, seed, F_ = (int(v) for v in g['cfg'])
    tr = hal['tr']
    enc = tr.AudioEncoder(head_dim=hd, heads=heads, layers=el, p_drop=0.2, input_dim=F_, conv_dim=conv_dim, conv_strides=strides)
    dec = tr.CTCAttentionDecoder(vocab=vocab, head_dim=hd, heads=heads, p_drop=0.2, layers=dl)
    enc.load_state_dict(pe, strict=True)
    dec.load_state_dict(pd, strict=True)
    return g, pd, batch, heads, enc.to(DEV).eval(), dec.to(DEV).eval()


# ---------------------------------------------------------------------------------------- operators
@pytest.mark.parametrize('hd,heads,N,Tq,Tk,causal,ragged', [
    (64, 3, 2, 130, 130, True, False), (64, 2, 3, 70, 70, False, True), (32, 4, 2, 9, 100, False, True),
    (16, 2, 3, 5, 5, True, False), (64, 1, 2, 200, 200, False, False), (32, 2, 2, 65, 65, True, False),
    (64, 2, 2, 1, 77, False, True)])
@pytest.mark.parametrize('math_mode', ['f32', 'bf16x3', 'bf16'], indirect=True)
def test_attention_fwd_against_sdpa(hal, hd, heads, N, Tq, Tk, causal, ragged, math_mode):
    g = torch.Generator().manual_seed(hd + Tq + Tk)
    C = heads * hd
    q = torch.randn(N * Tq, C, generator=g)
    kv = torch.randn(N * Tk, 2 * C, generator=g)                       # packed k|v rows, like the fused GEMM output
    lens = torch.tensor([Tk - (7 * n) % Tk for n in range(N)], dtype=torch.int32) if ragged else None
    y, lse, ent = hal['ops'].attention_fwd(q.to(DEV), kv[:, :C].to(DEV), kv[:, C:].to(DEV), N, heads, hd, Tq, Tk, causal=causal,
                                           key_lengths=lens.to(DEV) if ragged else None, want_lse=True, want_entropy=True)
    qh = q.view(N, Tq, heads, hd).transpose(1, 2)
    kh = kv[:, :C].reshape(N, Tk, heads, hd).transpose(1, 2)
    vh = kv[:, C:].reshape(N, Tk, heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(), float('-inf'))
    if ragged:
        s = s.masked_fill((torch.arange(Tk)[None, :] >= lens[:, None])[:, None, None, :], float('-inf'))
    att = s.softmax(-1)
    ref = (att @ vh).transpose(1, 2).reshape(N * Tq, C)
    # exact-f32 MFMA / split-bf16 (three passes, ~2^-16 per product) / plain bf16 operands (2^-9 per operand)
    tol = {'f32': 3e-6, 'bf16x3': 3e-5, 'bf16': 3e-2}[math_mode]
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=tol, rtol=1e-5)
    np.testing.assert_allclose(lse.cpu().numpy(), torch.logsumexp(s, -1).numpy(), atol=max(tol, 1e-5) * 4, rtol=1e-6)
    # the entropy monitor always runs on the exact-f32 kernel
    np.testing.assert_allclose(ent.cpu().numpy(), (-att * torch.log(att + 1e-8)).sum(-1).numpy(), atol=2e-5, rtol=1e-5)
    y2, _, _ = hal['ops'].attention_fwd(q.to(DEV), kv[:, :C].to(DEV), kv[:, C:].to(DEV), N, heads, hd, Tq, Tk, causal=causal,
                                        key_lengths=lens.to(DEV) if ragged else None)
    np.testing.assert_allclose(y2.cpu().numpy(), ref.numpy(), atol=tol, rtol=1e-5)


def test_rope_attend_lengths_match_reference(hal):
    tr = hal['tr']
    g = load_golden('g6_asr_parts')
    for nm in 'abc':
        y = tr.rotate_interleaved(torch.from_numpy(g[f'rope.{nm}.x']).to(DEV), t0=int(g[f'rope.{nm}.t0']))
        np.testing.assert_allclose(y.cpu().numpy(), g[f'rope.{nm}.y'], rtol=0, atol=2e-6)
    q, k, v, mask = (torch.from_numpy(g['attend.' + n]).to(DEV) for n in ('q', 'k', 'v', 'mask'))
    # the fixture's mask is not a suffix mask: the general kernel (halo_attention_masked), against the reference's own output
    y, ent = tr.attend(q, k, v, mask)
    np.testing.assert_allclose(y.cpu().numpy(), g['attend.y'], atol=2e-6)
    np.testing.assert_allclose(float(ent), float(g['attend.entropy']), rtol=1e-5)
    y, ent = tr.attend(q, k, v, None)
    from oracle import transformer_ref
    yr, er = transformer_ref.attend(q.cpu(), k.cpu(), v.cpu(), None)
    np.testing.assert_allclose(y.cpu().numpy(), yr.numpy(), atol=2e-6)
    np.testing.assert_allclose(float(ent), float(er), rtol=1e-5)
    lens = torch.from_numpy(g['lengths.in'])
    conv = hal['conv']
    assert np.array_equal(conv.ConvEncoder(input_dim=8, hidden_dim=8, output_dim=8, strides=(2, 2, 2)).subsampled_lengths(lens).numpy(),
                          g['lengths.s222'])
    assert np.array_equal(conv.ConvEncoder(input_dim=8, hidden_dim=8, output_dim=8, strides=(2, 2, 1)).subsampled_lengths(lens).numpy(),
                          g['lengths.s221'])


@BOTH_MODES
def test_conv_frontend_matches_oracle(hal, math_mode):
    from oracle import transformer_ref
    pe = transformer_ref.make_encoder_params(64, 8, 0, 80, 256, 3, 5)
    conv = hal['conv'].ConvEncoder(input_dim=80, hidden_dim=256, output_dim=512, strides=(2, 2, 2))
    conv.load_state_dict({k[len('conv.'):]: v for k, v in pe.items() if k.startswith('conv.')}, strict=True)
    conv = conv.to(DEV).eval()
    x = torch.randn(3, 80, 77, generator=torch.Generator().manual_seed(1))         # [N, F, T] like the reference
    lens = torch.tensor([77, 60, 9])
    with torch.no_grad():
        y, olen = conv(x.to(DEV), lens)
        ref = transformer_ref.conv_encoder(pe, 'conv.', x, (2, 2, 2))
    assert y.shape == ref.shape and olen.dtype == torch.int32
    assert olen.tolist() == transformer_ref.subsampled_lengths(lens, (2, 2, 2)).tolist()
    # bf16x3 products carry ~2^-16 relative error each; the activations here reach ~10
    np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=2e-5 if math_mode == 'f32' else 1e-4, rtol=1e-5)


# ------------------------------------------------------------------------------------------- models
@BOTH_MODES
@pytest.mark.parametrize('name', ASR_CASES)
def test_asr_encoder_decoder_matches_reference(hal, name, math_mode):
    g, pd, (x, il, tg, tl), heads, enc, dec = _models(hal, name)
    N = x.shape[0]
    deep = 'transformer32' in name
    with torch.no_grad():
        feats, flen, stats = enc(x.to(DEV), il.to(DEV))
        assert flen.dtype == torch.int32 and np.array_equal(flen.cpu().numpy(), g['feature_lengths'])
        np.testing.assert_allclose(feats.cpu().numpy(), g['features'], rtol=0, atol=2e-4 if deep else 2e-5)
        assert all(float(e) == float('-inf') for e in stats['meme_entropy'] + stats['self_entropy'])
        feats = torch.from_numpy(g['features']).to(DEV)                  # downstream checks start from the reference's features
        for red in ('mean', 'none', 'sumeach'):
            loss, _ = dec.decoder(feats, tg.to(DEV), flen, tl.to(DEV), reduction=red, drop_labels=False)
            np.testing.assert_allclose(loss.cpu().numpy(), g['decoder_loss.' + red], rtol=2e-5, atol=1e-4, err_msg=red)
        cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1)
        joint, _ = dec(feats, cond.to(DEV), flen, (tl + 1).to(DEV))
        np.testing.assert_allclose(float(joint), float(g['joint_loss']), rtol=2e-5)
        _, stats = dec.decoder(feats, tg.to(DEV), flen, tl.to(DEV), measure_entropy=True, drop_labels=False)
        np.testing.assert_allclose(np.array([float(e) for e in stats['meme_entropy']]), g['meme_entropy'], rtol=2e-4, atol=1e-5)
        np.testing.assert_allclose(np.array([float(e) for e in stats['self_entropy']]), g['self_entropy'], rtol=2e-4, atol=1e-5)


@BOTH_MODES
@pytest.mark.parametrize('name', ASR_CASES)
def test_asr_greedy_decode_matches_reference(hal, name, math_mode):
    from oracle import transformer_ref
    g, pd, (x, il, tg, tl), heads, enc, dec = _models(hal, name)
    feats, flen = torch.from_numpy(g['features']), torch.from_numpy(g['feature_lengths'])
    with torch.no_grad():
        outs, olen, ali, lps, ents = dec.decode(feats.to(DEV), flen.to(DEV), tl.to(DEV))
        o_outs, o_len, o_lps, o_ents, _ = transformer_ref.decoder_decode(pd, feats, flen, tl, heads, pre='decoder.')
    assert olen.dtype == torch.int32 and ali == [None] * len(tl)
    assert np.array_equal(olen.cpu().numpy(), g['decode.output_lengths'])
    assert [o.tolist() for o in outs.unbind()] == _unpad(g['decode.tokens'], g['decode.token_lens'])
    np.testing.assert_allclose(lps.cpu().numpy(), o_lps.numpy(), rtol=0, atol=2e-3)
    np.testing.assert_allclose(ents.cpu().numpy(), o_ents.numpy(), rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(lps.cpu().numpy(), g['decode.log_probs'], rtol=1e-2, atol=5e-2)
    np.testing.assert_allclose(ents.cpu().numpy(), g['decode.sum_entropies'], rtol=1e-2, atol=5e-2)
    with torch.no_grad():
        outs, olen, _, lps, _ = dec.decode(feats.to(DEV), flen.to(DEV), tl.to(DEV), prompt=torch.tensor([[7, 9]] * len(tl)))
    assert np.array_equal(olen.cpu().numpy(), g['decode_prompt.output_lengths'])
    assert [o.tolist() for o in outs.unbind()] == _unpad(g['decode_prompt.tokens'], g['decode_prompt.token_lens'])
    np.testing.assert_allclose(lps.cpu().numpy(), g['decode_prompt.log_probs'], rtol=1e-2, atol=5e-2)


def test_block_and_mha_public_forward(hal):
    """The module-level call surface (ha/transformer.py:289-300, 464-471): Block / MultiHeadAttention called
    directly with [N, T, C] tensors, the key-padding mask Block builds, and the refusals."""
    from oracle import transformer_ref
    tr = hal['tr']
    hd, heads, N, T, S = 16, 2, 2, 7, 9
    C = hd * heads
    p = {}
    transformer_ref._block_params(p, torch.Generator().manual_seed(9), '', C, memory=True)
    blk = tr.Block(head_dim=hd, heads=heads, p_drop=0.1, memory=True)
    blk.load_state_dict(p, strict=True)
    blk = blk.to(DEV).eval()
    g = torch.Generator().manual_seed(10)
    x, mem = torch.randn(N, T, C, generator=g), torch.randn(N, S, C, generator=g)
    mlen = torch.tensor([9, 4])
    ents = []
    with torch.no_grad():
        y, (m_ent, t_ent) = blk(x.to(DEV), causal=True, memory=mem.to(DEV), memory_lengths=mlen.to(DEV), measure_entropy=True)
        ref = transformer_ref.block(p, '', x, heads, causal=True, memory=mem, memory_lengths=mlen, entropies=ents)
        np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), atol=5e-6, rtol=1e-5)
        np.testing.assert_allclose([float(m_ent), float(t_ent)], [float(ents[0][0]), float(ents[0][1])], rtol=1e-5)
        mask = (torch.arange(S)[None, :] >= mlen[:, None])[:, None, None, :]
        xn = transformer_ref.layer_norm(x, p['ln_time.weight'])
        m, _ = blk.mix_memory(xn.to(DEV), mem.to(DEV), mask=mask.to(DEV))
        np.testing.assert_allclose(m.cpu().numpy(), transformer_ref.mha(p, 'mix_memory.', xn, mem, heads, key_mask=mask[:, 0, 0]).numpy(),
                                   atol=5e-6, rtol=1e-5)
        with pytest.raises(NotImplementedError):
            blk(x.to(DEV), memory=mem.to(DEV), memory_lengths=mlen.to(DEV), kv_cache_parts=tr.BlockKVCache(memory=(None,) * 4, time=None))
    with pytest.raises(NotImplementedError):
        blk(x.to(DEV), memory=mem.to(DEV), memory_lengths=mlen.to(DEV))       # grad mode: the backward is not built
    with pytest.raises(hal['lib'].HaloError):
        with torch.no_grad():
            blk(x, memory=mem, memory_lengths=mlen)                              # CPU tensors: no fallback


# ------------------------------------------------------------------------------------ training direction
def test_dwconv_backward_against_autograd(hal):
    g = torch.Generator().manual_seed(4)
    N, T, C, ks = 3, 37, 24, 3
    for stride in (1, 2, 3):
        x = torch.randn(N, T, C, generator=g, requires_grad=True)
        w = torch.randn(C, 1, ks, generator=g, requires_grad=True)
        b = torch.randn(C, generator=g, requires_grad=True)
        y = F.conv1d(x.mT, w, b, stride=stride, padding=1, groups=C).mT
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        dx, dw, db = hal['ops'].dwconv1d_cl_bwd(dy.contiguous().to(DEV), x.detach().to(DEV), w.detach().view(C, ks).to(DEV), stride, 1)
        np.testing.assert_allclose(dx.cpu().numpy(), x.grad.numpy(), atol=1e-5)
        np.testing.assert_allclose(dw.cpu().numpy(), w.grad.view(C, ks).numpy(), atol=5e-5)
        np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), atol=5e-5)


def _joint_backward(hal, name):
    g, pd, (x, il, tg, tl), heads, enc, dec = _models(hal, name)
    cond = torch.cat([torch.full((x.shape[0], 1), 5, dtype=torch.long), tg], dim=1)
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))                    # eval mode + grad: no dropout, like the fixture
    assert feats.requires_grad
    joint, _ = dec(feats, cond.to(DEV), flen, (tl + 1).to(DEV))
    joint.backward()
    return g, enc, dec, joint


@BOTH_MODES
@pytest.mark.parametrize('name', ['g6_asr_tiny', 'g6_asr_tiny_s221', 'g6_asr_tiny_stop'])
def test_asr_tiny_gradients_match_reference(hal, name, math_mode):
    """loss.backward() of encoder -> (decoder CE + 0.3 CTC) through the HIP path vs the reference's gradients: every parameter."""
    g, enc, dec, joint = _joint_backward(hal, name)
    np.testing.assert_allclose(joint.item(), float(g['train.joint_loss']), rtol=2e-5)
    n = 0
    for pre, m in (('encoder.', enc), ('decoder.', dec)):
        for k, p in m.named_parameters():
            want = g['grad.' + pre + k]
            scale = max(1.0, float(np.abs(want).max()))
            tol = dict(rtol=2e-3, atol=2e-5 * scale) if math_mode == 'f32' else dict(rtol=5e-3, atol=1e-4 * scale)
            np.testing.assert_allclose(p.grad.cpu().numpy(), want, err_msg=pre + k, **tol)
            n += 1
    assert n == sum(1 for k in g if k.startswith('grad.'))


@BOTH_MODES
def test_asr_transformer32_gradients_match_reference(hal, math_mode):
    """`transformer:32` (12+12 layers): norm and a strided sample of every parameter's gradient vs the reference's backward."""
    g, enc, dec, joint = _joint_backward(hal, 'g6_asr_transformer32')
    np.testing.assert_allclose(joint.item(), float(g['train.joint_loss']), rtol=5e-5)
    rel = 5e-3 if math_mode == 'f32' else 1e-2
    for pre, m in (('encoder.', enc), ('decoder.', dec)):
        for k, p in m.named_parameters():
            want_norm = float(g['gradnorm.' + pre + k])
            assert abs(float(p.grad.norm()) - want_norm) <= rel * want_norm + 1e-7, pre + k
            sample = p.grad.flatten()[::max(1, p.numel() // 500)][:500].cpu().numpy()
            want = g['gradsample.' + pre + k]
            assert np.abs(sample - want).max() <= rel * max(np.abs(want).max(), want_norm / p.numel() ** 0.5) + 1e-7, pre + k


def test_asr_training_mode_refusals(hal):
    tr = hal['tr']
    dec = tr.Decoder(vocab=16, head_dim=16, heads=2, p_drop=0.2, layers=1).to(DEV).train()
    feats = torch.randn(2, 5, 32, device=DEV)
    tg = torch.randint(4, 16, (2, 3), device=DEV)
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            dec(feats, tg, torch.tensor([5, 4], device=DEV), torch.tensor([3, 2], device=DEV))  # train-mode dropout without autograd
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            dec.h[0](feats, memory=feats, memory_lengths=torch.tensor([5, 4], device=DEV))       # bare Block in training mode
    loss, _ = dec(feats, tg, torch.tensor([5, 4], device=DEV), torch.tensor([3, 2], device=DEV))  # label dropout + dropout: trains
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in dec.parameters())


def _block_masks(philox, seed, offset, p, first_site, N, heads, T, S, C, memory):
    """The multipliers of one Block's dropout sites (forward order, see Block._forward2d_train) for the CPU oracle."""
    sid, mk = first_site, {}
    def rows(n_rows):
        nonlocal sid
        m = torch.from_numpy(philox.dropout_mask(n_rows * C, p, seed, sid, offset)).view(N, n_rows // N, C)
        sid += 1
        return m
    def att(Tq, Tk):
        nonlocal sid
        m = torch.from_numpy(philox.attention_dropout_mask(N, heads, Tq, Tk, p, seed, sid, offset).copy())
        sid += 1
        return m
    if memory:
        mk['cross_att'] = att(T, S)
        mk['cross_out'] = rows(N * T)
    mk['self_att'] = att(T, T)
    mk['self_out'] = rows(N * T)
    mk['mlp_out'] = rows(N * T)
    return mk, sid


@BOTH_MODES
@pytest.mark.parametrize('name', ['g6_asr_tiny', 'g6_asr_tiny_s221'])
def test_asr_training_mode_dropout_matches_oracle_with_same_masks(hal, name, math_mode):
    """model.train() with p_drop = 0.2: the Philox masks of every dropout site (encoder input, attention probabilities in the
    kernels, proj / MLP outputs in the GEMM epilogues) are restated on the CPU and fed to the oracle; loss and every
    gradient must agree like in eval mode."""
    from oracle import philox, transformer_ref
    g, pe, pd, (x, il, tg, tl), heads, strides = asr_case_from_golden(name)
    vocab, hd, heads, el, dl, conv_dim, N, T, S_t, seed0, F_ = (int(v) for v in g['cfg'])
    tr = hal['tr']
    P, SEED = 0.2, 0x1234567890ABCDEF
    enc = tr.AudioEncoder(head_dim=hd, heads=heads, layers=el, p_drop=P, input_dim=F_, conv_dim=conv_dim, conv_strides=strides)
    dec = tr.Decoder(vocab=vocab, head_dim=hd, heads=heads, p_drop=P, layers=dl)
    enc.load_state_dict(pe, strict=True)
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in pd.items() if k.startswith('decoder.')}, strict=True)
    enc, dec = enc.to(DEV).train(), dec.to(DEV).train()
    enc.dropout_stream.seed = dec.dropout_stream.seed = SEED
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))
    loss, _ = dec(feats, tg.to(DEV), flen, tl.to(DEV), drop_labels=False)
    loss.backward()
    # CPU oracle with the same masks
    C = hd * heads
    Tp = feats.shape[1]
    pe_r = {k: v.clone().requires_grad_(True) for k, v in pe.items()}
    pd_r = {k[len('decoder.'):]: v.clone().requires_grad_(True) for k, v in pd.items() if k.startswith('decoder.')}
    in_mult = torch.from_numpy(philox.dropout_mask(N * Tp * C, P, SEED, 64, 0)).view(N, Tp, C)
    sid, enc_masks = 65, []
    for _ in range(el):
        mk, sid = _block_masks(philox, SEED, 0, P, sid, N, heads, Tp, 0, C, memory=False)
        enc_masks.append(mk)
    Td = tg.shape[1] + 1
    sid, dec_masks = 64, []
    for _ in range(dl):
        mk, sid = _block_masks(philox, SEED, 0, P, sid, N, heads, Td, Tp, C, memory=True)
        dec_masks.append(mk)
    f_ref, fl_ref = transformer_ref.audio_encoder_forward(pe_r, x, il, heads, strides, in_mult=in_mult, block_masks=enc_masks)
    l_ref = transformer_ref.decoder_forward(pd_r, f_ref, tg, fl_ref, tl, heads, block_masks=dec_masks)
    l_ref.backward()
    np.testing.assert_allclose(feats.detach().cpu().numpy(), f_ref.detach().numpy(), atol=5e-5, rtol=1e-4)
    np.testing.assert_allclose(loss.item(), l_ref.item(), rtol=5e-5)
    for m, params in ((enc, pe_r), (dec, pd_r)):
        for k, p in m.named_parameters():
            want = params[k].grad.numpy()
            scale = max(1.0, float(np.abs(want).max()))
            tol = dict(rtol=2e-3, atol=2e-5 * scale) if math_mode == 'f32' else dict(rtol=5e-3, atol=1e-4 * scale)
            np.testing.assert_allclose(p.grad.cpu().numpy(), want, err_msg=k, **tol)
    # a second forward draws new masks (the stream offset advanced)
    feats2, _, _ = enc(x.to(DEV), il.to(DEV))
    assert not torch.allclose(feats2, feats)


# ------------------------------------------------------------------------------------- other shapes of the same path
@BOTH_MODES
@pytest.mark.parametrize('hd,heads,el,dl,conv_dim,N,T,S,strides,F_', [
    (16, 5, 1, 1, 40, 1, 300, 20, (2, 2, 1), 23),      # C = 80: not a multiple of 32 (LayerNorm-image fallback, ragged k-tiles), N = 1
    (64, 2, 2, 1, 64, 5, 531, 70, (2, 2, 2), 80),      # 67 encoder frames and 71 decoder positions: more than one 64-row attention tile
    (32, 3, 1, 2, 96, 2, 97, 3, (2, 1), 17),           # two convs only, C = 96
])
def test_asr_shape_robustness_against_oracle(hal, math_mode, hd, heads, el, dl, conv_dim, N, T, S, strides, F_):
    """Forward, losses and every gradient against the CPU oracle at sizes the goldens do not cover (no fixture: the oracle is
    pinned to the reference by tests/test_oracle_golden.py)."""
    from oracle import transformer_ref
    vocab = 29
    pe = transformer_ref.make_encoder_params(hd, heads, el, F_, conv_dim, len(strides), 3)
    pd = transformer_ref.make_decoder_params(vocab, hd, heads, dl, 4)
    x, il, tg, tl = transformer_ref.synthetic_asr_batch(N, T, F_, vocab, S, 5)
    tr = hal['tr']
    enc = tr.AudioEncoder(head_dim=hd, heads=heads, layers=el, p_drop=0.1, input_dim=F_, conv_dim=conv_dim, conv_strides=strides)
    dec = tr.CTCAttentionDecoder(vocab=vocab, head_dim=hd, heads=heads, p_drop=0.1, layers=dl)
    enc.load_state_dict(pe, strict=True); dec.load_state_dict(pd, strict=True)
    enc, dec = enc.to(DEV).eval(), dec.to(DEV).eval()
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))
    loss, _ = dec.decoder(feats, tg.to(DEV), flen, tl.to(DEV), drop_labels=False)
    loss.backward()
    pe_r = {k: v.clone().requires_grad_(True) for k, v in pe.items()}
    pd_r = {k[len('decoder.'):]: v.clone().requires_grad_(True) for k, v in pd.items() if k.startswith('decoder.')}
    f_ref, fl_ref = transformer_ref.audio_encoder_forward(pe_r, x, il, heads, strides)
    l_ref = transformer_ref.decoder_forward(pd_r, f_ref, tg, fl_ref, tl, heads)
    l_ref.backward()
    assert np.array_equal(flen.cpu().numpy(), fl_ref.numpy())
    np.testing.assert_allclose(feats.detach().cpu().numpy(), f_ref.detach().numpy(), atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(loss.item(), l_ref.item(), rtol=5e-5)
    for m, params in ((enc, pe_r), (dec.decoder, pd_r)):
        for k, p in m.named_parameters():
            want = params[k].grad.numpy()
            scale = max(1.0, float(np.abs(want).max()))
            tol = dict(rtol=2e-3, atol=3e-5 * scale) if math_mode == 'f32' else dict(rtol=5e-3, atol=2e-4 * scale)
            np.testing.assert_allclose(p.grad.cpu().numpy(), want, err_msg=k, **tol)
    with torch.no_grad():                                      # greedy decode runs at these sizes and stops within the step budget
        outs, olen, _, lps, _ = dec.decode(feats.detach(), flen, tl.to(DEV))
    assert olen.shape == (N,) and int(olen.max()) <= int(tl.max()) + 1 and torch.isfinite(lps).all()


@pytest.mark.parametrize('T,causal,ragged', [(512, True, False), (520, True, False), (520, False, True), (640, True, True)])
def test_attention_two_query_blocks_per_wave_against_exact_kernel(hal, T, causal, ragged):
    """Shapes large enough that the single-pass forward runs 128-query workgroups (two 16-query blocks per wave, causal tiles
    paired long + short): against the exact-f32 kernels of the same library at the bf16 tolerance, forward and backward."""
    from haloop_amd import _lib
    ops = hal['ops']
    N, heads, hd = 4, 32, 64
    C = heads * hd
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(N * T, 3 * C, generator=g).to(DEV)
    dy = torch.randn(N * T, C, generator=g).to(DEV)
    lens = torch.tensor([T - (37 * n) % T for n in range(N)], dtype=torch.int32).to(DEV) if ragged else None
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    prev = _lib.get_math_mode()
    out = {}
    try:
        for mode in ('f32', 'bf16'):
            _lib.set_math_mode(mode)
            y, lse, _ = ops.attention_fwd(q, k, v, N, heads, hd, T, T, causal=causal, key_lengths=lens, want_lse=True)
            dqkv = torch.empty_like(qkv)
            ops.attention_bwd(q, k, v, y, dy, lse, dqkv[:, :C], dqkv[:, C:2 * C], dqkv[:, 2 * C:], N, heads, hd, T, T, causal=causal,
                              key_lengths=lens)
            out[mode] = (y.cpu().numpy(), lse.cpu().numpy(), dqkv.cpu().numpy())
    finally:
        _lib.set_math_mode(prev)
    np.testing.assert_allclose(out['bf16'][0], out['f32'][0], atol=3e-2, rtol=1e-2)
    np.testing.assert_allclose(out['bf16'][1], out['f32'][1], atol=5e-2, rtol=1e-3)
    scale = float(np.abs(out['f32'][2]).max())
    np.testing.assert_allclose(out['bf16'][2], out['f32'][2], atol=2e-2 * scale, rtol=0)


def test_graphed_train_step_equals_eager(hal):
    """haloop_amd.train.GraphedTrainStep: forward + backward of the ASR joint loss (dropout 0.2 on) replayed from one HIP graph gives
    the gradients of the eager launches bit for bit when the device dropout counter holds the same value, draws fresh masks on the
    next replay, and re-captures on a new input shape."""
    from oracle import transformer_ref
    from haloop_amd import train
    tr, lib = hal['tr'], hal['lib']
    prev = lib.get_math_mode()
    lib.set_math_mode('bf16x3')
    try:
        V, hd, heads, L, N, T, S = 32, 64, 2, 2, 6, 160, 5
        pe = transformer_ref.make_encoder_params(hd, heads, L, 80, 64, 3, 41)
        pd = transformer_ref.make_decoder_params(V, hd, heads, L, 42)
        enc = tr.AudioEncoder(head_dim=hd, heads=heads, layers=L, p_drop=0.2, input_dim=80, conv_dim=64)
        dec = tr.CTCAttentionDecoder(vocab=V, head_dim=hd, heads=heads, p_drop=0.2, layers=L)
        enc.load_state_dict(pe); dec.load_state_dict(pd)
        enc.to(DEV).train(); dec.to(DEV).train()
        x, il, tg, tl = transformer_ref.synthetic_asr_batch(N, T, 80, V, S, 7, ragged=True)
        cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1).to(DEV)
        xd, ild, tl1 = x.to(DEV), il.to(DEV), (tl + 1).to(DEV)
        params = list(enc.parameters()) + list(dec.parameters())

        def fwd(xd, ild, cond, tl1):
            f, fl, _ = enc(xd, ild)
            loss, _ = dec(f, cond, fl, tl1, drop_labels=False)          # label dropout draws from torch's generator: off for the comparison
            return loss

        step = train.GraphedTrainStep(fwd, params, dropout_streams=[enc.dropout_stream, dec.decoder.dropout_stream, dec.recognizer.dropout_stream])
        loss_g = step.step(xd, ild, cond, tl1)
        used = int(step.counter.item()) - 1                              # the counter value the replay drew its masks with
        grads_g = [p.grad.clone() for p in params]
        loss_g2 = step.step(xd, ild, cond, tl1)
        assert float(loss_g2) != float(loss_g)                           # the next replay: other dropout masks
        step.counter.fill_(used)
        for p in params:
            p.grad = None
        loss_e = fwd(xd, ild, cond, tl1)
        loss_e.backward()
        assert float(loss_e.detach()) == float(loss_g)
        for (name, p), gg in zip(list(enc.named_parameters()) + list(dec.named_parameters()), grads_g):
            if name.endswith('wte.weight'):      # the embedding gradient is a float atomic scatter-add: its order varies run to run
                np.testing.assert_allclose(p.grad.cpu().numpy(), gg.cpu().numpy(), rtol=0, atol=1e-6, err_msg=name)
            else:
                assert torch.equal(p.grad, gg), name
        loss_s = step.step(xd[:4], ild[:4], cond[:4], tl1[:4])           # a new shape re-captures
        assert torch.isfinite(loss_s) and params[0].grad is not None
    finally:
        lib.set_math_mode(prev)


def test_graphed_train_step_sees_optimizer_updates(hal):
    """A replay after an in-place optimizer step multiplies by the UPDATED weights: the GEMM operand images of the weights are built
    by launches inside the graph, not cached from the warm-up (a host-side cache hit records no launch, and the graph would go on
    reading the warm-up's images).  step(), update every parameter in place, step(): loss and every gradient equal the eager
    launches' on the updated weights bit for bit."""
    from oracle import transformer_ref
    from haloop_amd import train
    tr, lib = hal['tr'], hal['lib']
    prev = lib.get_math_mode()
    lib.set_math_mode('bf16x3')
    try:
        V, hd, heads, L, N, T, S = 32, 64, 2, 2, 6, 160, 5
        pe = transformer_ref.make_encoder_params(hd, heads, L, 80, 64, 3, 43)
        pd = transformer_ref.make_decoder_params(V, hd, heads, L, 44)
        enc = tr.AudioEncoder(head_dim=hd, heads=heads, layers=L, p_drop=0.0, input_dim=80, conv_dim=64)
        dec = tr.CTCAttentionDecoder(vocab=V, head_dim=hd, heads=heads, p_drop=0.0, layers=L)
        enc.load_state_dict(pe); dec.load_state_dict(pd)
        enc.to(DEV).train(); dec.to(DEV).train()
        dec.recognizer.dropout.p = 0.0                       # the CTC head's own dropout (recognizer.py:41) off too: no masks in this test
        x, il, tg, tl = transformer_ref.synthetic_asr_batch(N, T, 80, V, S, 9, ragged=True)
        cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1).to(DEV)
        xd, ild, tl1 = x.to(DEV), il.to(DEV), (tl + 1).to(DEV)
        named = list(enc.named_parameters()) + list(dec.named_parameters())
        params = [p for _, p in named]

        def fwd(xd, ild, cond, tl1):
            f, fl, _ = enc(xd, ild)
            loss, _ = dec(f, cond, fl, tl1, drop_labels=False)
            return loss

        step = train.GraphedTrainStep(fwd, params)
        loss0 = float(step.step(xd, ild, cond, tl1))
        with torch.no_grad():
            for p in params:                                 # a plain SGD step, in place like every optimizer of this package
                p.add_(p.grad, alpha=-0.5)
        loss_g = float(step.step(xd, ild, cond, tl1))
        assert loss_g != loss0
        grads_g = [p.grad.clone() for p in params]
        for p in params:
            p.grad = None
        loss_e = fwd(xd, ild, cond, tl1)
        loss_e.backward()
        assert float(loss_e.detach()) == loss_g
        for (name, p), gg in zip(named, grads_g):
            if name.endswith('wte.weight'):
                np.testing.assert_allclose(p.grad.cpu().numpy(), gg.cpu().numpy(), rtol=0, atol=1e-6, err_msg=name)
            else:
                assert torch.equal(p.grad, gg), name
    finally:
        lib.set_math_mode(prev)


def _edit_distance(a, b):
    d = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        prev, d[0] = d[0], i
        for j, cb in enumerate(b, 1):
            prev, d[j] = d[j], min(d[j] + 1, d[j - 1] + 1, prev + (ca != cb))
    return d[-1]


@pytest.mark.parametrize('math_mode', ['bf16x3', 'f32'], indirect=True)
def test_config5_composed_encoder_ctc_beam16_wer_is_zero(hal, math_mode):
    """BASELINE config 5 end to end on the `transformer:32` fixture inputs: AudioEncoder features -> the CTC head's log-probs ->
    ha.beam prefix search with beam 16 (haloop_amd.beam.decode_batch) -> hypotheses; against the CPU restatement of the same chain
    (oracle/transformer_ref.py + oracle/lattice.py, ha/transformer.py:202-258, ha/recognizer.py:43-46, ha/beam.py:71-137):
    word error rate 0 over all utterances for the top hypothesis, token-exact for every beam whose score is not tied."""
    from oracle import transformer_ref, lattice
    from haloop_amd import beam
    name = 'g6_asr_transformer32'
    g, pd, (x, il, tg, tl), heads, enc, dec = _models(hal, name)
    _, pe, _, _, _, strides = asr_case_from_golden(name)
    with torch.no_grad():
        feats, flen, _ = enc(x.to(DEV), il.to(DEV))
        lp = dec.recognizer.log_probs(feats)
        hyps, scores = beam.decode_batch(lp, beam_size=16)
        f_ref, fl_ref = transformer_ref.audio_encoder_forward(pe, x, il, heads, strides)
        lp_ref = F.linear(f_ref, pd['recognizer.classifier.weight'], pd['recognizer.classifier.bias']).log_softmax(-1)
    assert np.array_equal(flen.cpu().numpy(), fl_ref.numpy())
    errs = words = 0
    for n in range(x.shape[0]):
        ref_hyps, ref_scores = lattice.ctc_beam_search_decode_logits(lp_ref[n], 16)
        errs += _edit_distance(hyps[n][0], ref_hyps[0]); words += len(ref_hyps[0])
        np.testing.assert_allclose(scores[n].cpu().numpy(), np.asarray(ref_scores, dtype=np.float32), rtol=0, atol=5e-3)
        rs = np.asarray(ref_scores, dtype=np.float64)
        for k in range(16):          # a beam whose score is separated from its neighbours by more than the feature tolerance is token-exact
            gap = min(abs(rs[k] - rs[j]) for j in range(16) if j != k)
            if gap > 2e-2:
                assert hyps[n][k] == list(ref_hyps[k]), (n, k)
    assert errs == 0 and words > 0, (errs, words)
