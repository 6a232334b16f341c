"""BASELINE config 2 on the GPU (LC-2x1024 at B=64: the benchmarked grid, 64 hidden tiles x 4 batch tiles = 256 workgroups) and the
weight-resident persistent recurrence (csrc/lstm_persist.hip) against the reference-generated fixtures, the CPU oracle and the
per-step launch chain it replaces.  Tolerances are the fp32-grade ones of tests/test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def hal():
    from haloop_amd import _lib, ops, rnn, recognizer
    _lib.lib()
    _lib.lend_scratch()
    return dict(ops=ops, rnn=rnn, recognizer=recognizer, lib=_lib)


@pytest.fixture
def math_mode(request, hal):
    prev = hal['lib'].get_math_mode()
    hal['lib'].set_math_mode(request.param)
    yield request.param
    hal['lib'].set_math_mode(prev)


@pytest.fixture
def persistent(request, hal):
    hal['lib'].set_lstm_persistent(request.param)
    yield request.param
    hal['lib'].set_lstm_persistent(True)


PERSIST = pytest.mark.parametrize('persistent', [True, False], indirect=True)


def _status(hal, buf, backward, T, B, in0, H, L):
    off = hal['lib'].lib().halo_lstm_status_offset(int(backward), T, B, in0, H, L)
    return int(buf.view(torch.int32)[off // 4].item())


@PERSIST
@pytest.mark.parametrize('math_mode', ['f32', 'bf16x3'], indirect=True)
def test_lc2x1024_b64_matches_reference(hal, math_mode, persistent):
    """Config 2's real grid against the reference's own numbers (fixture g1_lc2x1024_b64, ragged lengths)."""
    from oracle import cpu_ref
    g = load_golden('g1_lc2x1024_b64')
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    enc_p, rec_p = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], c['seed'])
    x, _, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], c['seed'])
    enc = hal['rnn'].Encoder(c['F_'], c['C'], c['H'], num_layers=c['L'])
    rec = hal['recognizer'].TemporalClassifier(c['H'], c['V'])
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    il = torch.from_numpy(g['il']).to(DEV)
    feats, flen, _ = enc(x.to(DEV), il)
    feats.retain_grad()
    loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=1e-5)
    np.testing.assert_allclose(feats[:, :, ::61].detach().cpu().numpy(), g['feats_slice'], atol=1e-4)
    np.testing.assert_allclose(feats.detach().double().sum().item(), float(g['feats_sum']), rtol=1e-5)
    assert np.array_equal(flen.cpu().numpy(), g['flen'])
    loss.backward()
    np.testing.assert_allclose(feats.grad[:, :, ::61].cpu().numpy(), g['dfeats_slice'], atol=1e-6)
    for k, p in list(enc.named_parameters()) + list(rec.named_parameters()):
        key = ('recognizer.' if k.startswith('classifier') else 'encoder.') + k
        np.testing.assert_allclose(p.grad.double().norm().item(), float(g['gradnorm.' + key]), rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(p.grad.reshape(-1)[::9973].cpu().numpy(), g['gradslice.' + key], rtol=1e-3, atol=1e-6, err_msg=key)
    with torch.no_grad():
        lp = rec.log_probs(feats)
    np.testing.assert_allclose(lp[::3].cpu().numpy(), g['lp_slice'], atol=1e-4)
    ali, scores, hyp, hlen = hal['ops'].ctc_greedy(lp.contiguous())
    # greedy alignments of a random-init model: frames whose two best log-probs are closer than the feature tolerance may flip
    lp_ref_top2 = np.sort(g['lp_slice'], axis=-1)[..., -2:]
    decisive = (lp_ref_top2[..., 1] - lp_ref_top2[..., 0]) > 1e-4          # the log-probs themselves are held to 1e-4 above
    assert decisive.mean() >= 0.95, decisive.mean()              # the filter may drop near-ties only, never most of the test
    assert np.array_equal(ali.cpu().numpy()[::3][decisive], g['ali'][::3][decisive])
    # utterances (of the sampled third) without a single near-tie: collapsed hypotheses and their lengths are exact as well
    sure = np.nonzero(decisive.all(axis=1))[0] * 3
    assert len(sure) >= 5
    assert np.array_equal(hlen.cpu().numpy()[sure], g['hlen'][sure])
    for n in sure:
        k = int(g['hlen'][n])
        assert np.array_equal(hyp[n, :k].cpu().numpy(), g['hyps'][n, :k]), n
    if persistent and math_mode != 'f32':
        assert hal['lib'].lib().halo_lstm_persistent_eligible(c['B'], c['H']) == 1


@pytest.mark.parametrize('math_mode', ['f32', 'bf16x3'], indirect=True)
@pytest.mark.parametrize('use_graph', [True, False])
def test_three_train_steps_b64_match_reference(hal, math_mode, use_graph):
    """Three optimizer steps at config 2's shape against the reference's own run (fixture g1_train3_b64)."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    g = load_golden('g1_train3_b64')
    c = {k[4:]: v for k, v in g.items() if k.startswith('cfg_')}
    F_, C, H, L, V = (int(c[k]) for k in ('F_', 'C', 'H', 'L', 'V'))
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, int(c['seed']))
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    tr = LstmCtcTrainer(enc, rec, lr=float(c['lr']), use_graph=use_graph)
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(int(c['B']), int(c['T']), F_, V, int(c['S']), 200 + step)
        loss = tr.step(x.to(DEV), il.to(DEV), tg.to(DEV), tl.to(DEV))
        np.testing.assert_allclose(loss.item(), g['losses'][step], rtol=2e-5)
        np.testing.assert_allclose(tr.grad_norm.item(), g['gnorms'][step], rtol=1e-4)
    sd = {**{'encoder.' + k: v for k, v in enc.state_dict().items()}, **{'recognizer.' + k: v for k, v in rec.state_dict().items()}}
    atol = 5e-6 if math_mode == 'f32' else 2e-5          # same bound as test_train_steps_match_reference
    for k, v in sd.items():
        np.testing.assert_allclose(v.reshape(-1)[::4999].cpu().numpy(), g['finalslice.' + k], atol=atol, err_msg=k)


def _lstm_case(hal, T, B, in0, H, L, p_drop, seed, with_state):
    ops = hal['ops']
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, B, in0, generator=g).to(DEV)
    k = 1.0 / H ** 0.5
    w_ih = [((torch.rand(4 * H, in0 if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_ih = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_hh = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    h0 = (torch.randn(L, B, H, generator=g) * 0.3).to(DEV) if with_state else None
    c0 = (torch.randn(L, B, H, generator=g) * 0.3).to(DEV) if with_state else None
    dy = torch.randn(T, B, H, generator=g).to(DEV)
    dhn = (torch.randn(L, B, H, generator=g) * 0.1).to(DEV) if with_state else None
    dcn = (torch.randn(L, B, H, generator=g) * 0.1).to(DEV) if with_state else None
    drop = ops.Dropout(p_drop, 1234, 3) if p_drop > 0 else ops.NO_DROPOUT
    y, hn, cn, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, h0=h0, c0=c0, want_state=True, drop=drop)
    st_f = _status(hal, reserve, False, T, B, in0, H, L)
    ws = ops.lstm_bwd_workspace(x, w_hh)
    dx, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, dhn=dhn, dcn=dcn, want_dx=True, drop=drop, workspace=ws)
    st_b = _status(hal, ws, True, T, B, in0, H, L)
    out = {'y': y, 'hn': hn, 'cn': cn, 'dx': dx}
    for name, lst in grads.items():
        for l, t in enumerate(lst):
            out[f'{name}{l}'] = t
    return {k: v.cpu() for k, v in out.items()}, (st_f, st_b)


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,L,p_drop,with_state', [
    (7, 5, 40, 256, 2, 0.0, True),          # 16 workgroups, one batch tile, carried state
    (5, 33, 64, 512, 1, 0.0, False),        # 96 workgroups, 3 batch tiles (falls back to the plain block map)
    (6, 16, 128, 768, 2, 0.25, False),      # 48 workgroups, inter-layer dropout
    (21, 64, 128, 1024, 2, 0.2, False),     # the benchmark's grid: 256 workgroups
    (3, 1, 32, 1024, 1, 0.0, True),         # a single utterance
])
def test_persistent_recurrence_equals_step_chain(hal, math_mode, T, B, in0, H, L, p_drop, with_state):
    """The one-launch recurrence and the T-launch chain compute the same function; only the order of the K-slice sums differs."""
    assert hal['lib'].lib().halo_lstm_persistent_eligible(B, H) == 1
    hal['lib'].set_lstm_persistent(True)
    a, st_a = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    hal['lib'].set_lstm_persistent(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    finally:
        hal['lib'].set_lstm_persistent(True)
    assert st_a == (0, 0) and st_b == (0, 0)                       # no bounded wait timed out
    tol = dict(rtol=2e-4, atol=2e-5) if math_mode == 'bf16x3' else dict(rtol=3e-2, atol=3e-3)
    for k in a:
        scale = float(b[k].abs().max()) + 1e-12
        np.testing.assert_allclose(a[k].numpy() / scale, b[k].numpy() / scale, err_msg=k, **tol)


@pytest.mark.parametrize('T,B,in0,H,L,p_drop', [
    (21, 64, 128, 1024, 2, 0.2),            # the benchmark's grid; T*B = 1344 rows: the last 128-row tile of the row image is half padding
    (8, 32, 64, 256, 2, 0.0),               # T*B = 256: whole tiles, one batch-tile pair
    (5, 96, 128, 512, 1, 0.0),              # 3 x 2 batch tiles (plain block map), a single layer with dx
])
@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)     # bf16: the hi parts only
def test_backward_chain_writes_the_same_operand_images(hal, math_mode, T, B, in0, H, L, p_drop):
    """The persistent backward writes the split-bf16 GEMM operand images of the gate gradients itself (B % 32 == 0) instead of leaving
    them to the operand-image launch: the bits must be the same, so every weight and input gradient is identical (the bias gradients are
    then summed inside the chain too, in another order: fp32 rounding apart)."""
    assert hal['lib'].lib().halo_lstm_persistent_eligible(B, H) == 1
    hal['lib'].set_lstm_persistent_images(True)
    a, st_a = _lstm_case(hal, T, B, in0, H, L, p_drop, 6, False)
    hal['lib'].set_lstm_persistent_images(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, L, p_drop, 6, False)
    finally:
        hal['lib'].set_lstm_persistent_images(True)
    assert st_a == (0, 0) and st_b == (0, 0)
    for k in a:
        if k.startswith('db'):      # the bias gradients: summed per batch tile over time in the chain, then over the tiles -- another order
            scale = float(b[k].abs().max()) + 1e-12
            np.testing.assert_allclose(a[k].numpy() / scale, b[k].numpy() / scale, rtol=0, atol=2e-6, err_msg=k)
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('math_mode', ['bf16x3'], indirect=True)
def test_persistent_recurrence_sees_fresh_data_on_every_launch(hal, math_mode):
    """Hand-off buffers are re-used across launches (torch's allocator returns the same reserve): results must follow the inputs
    of THIS launch, never lines cached from the previous one.  Runs the forward on alternating inputs and compares each result
    with the step-chain result for the same input."""
    T, B, in0, H, L = 21, 64, 128, 1024, 1
    ops = hal['ops']
    g = torch.Generator().manual_seed(9)
    k = 1.0 / H ** 0.5
    w_ih = [((torch.rand(4 * H, in0, generator=g) * 2 - 1) * k).to(DEV)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV)]
    b = [torch.zeros(4 * H, device=DEV)]
    xs = [torch.randn(T, B, in0, generator=g).to(DEV) * s for s in (1.0, -0.5, 2.0)]
    hal['lib'].set_lstm_persistent(False)
    try:
        refs = [ops.lstm_fwd(x, w_ih, w_hh, b, b)[0].cpu() for x in xs]
    finally:
        hal['lib'].set_lstm_persistent(True)
    for rep in range(4):
        for x, ref in zip(xs, refs):
            y = ops.lstm_fwd(x, w_ih, w_hh, b, b)[0]
            np.testing.assert_allclose(y.cpu().numpy(), ref.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize('B,T,C,p', [(64, 79, 512, 0.0), (64, 79, 512, 0.2), (5, 43, 64, 0.2), (32, 8, 128, 0.0)])
def test_fused_front_end_launch(B, T, C, p):
    """The LC front-end shape (F = 80, kernel 5 -> K = 400) runs as ONE launch (csrc/elementwise.hip, subsample_fused_kernel: im2col
    tile in LDS, fp32 MFMA, bias + relu + dropout): against torch's float64 conv1d (ha/rnn.py:16-19) with the oracle's Philox mask,
    and the saved im2col image against unfold -- ragged last row tiles ((T' * B) % 16 != 0) and zero-padded frames at both ends."""
    from haloop_amd import _lib, ops
    from haloop_amd.ops import Dropout, NO_DROPOUT
    from oracle import philox
    from oracle.cpu_ref import STREAM_SUBSAMPLE
    _lib.lib()
    F_, ks, stride, pad = 80, 5, 4, 3
    g = torch.Generator().manual_seed(B + T)
    x = torch.randn(B, T, F_, generator=g)
    w = torch.randn(C, F_, ks, generator=g) / 20
    bias = torch.randn(C, generator=g)
    seed, offset = 1234, 7
    y, col = ops.subsample_fwd(x.cuda(), w.cuda(), bias.cuda(), Dropout(p, seed, offset, None) if p else NO_DROPOUT)
    Tp = ops.subsampled_length(T, ks, stride, pad)
    ref = torch.relu(torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=stride, padding=pad))
    ref = ref.permute(2, 0, 1)                                                          # [T', B, C]
    if p:
        mask = torch.from_numpy(philox.dropout_mask(Tp * B * C, p, seed, STREAM_SUBSAMPLE, offset)).view(Tp, B, C)
        ref = ref * mask.double()
    assert y.shape == (Tp, B, C)
    np.testing.assert_allclose(y.cpu().double().numpy(), ref.numpy(), rtol=0, atol=2e-6 * float(ref.abs().max()))
    unf = torch.nn.functional.unfold(torch.nn.functional.pad(x.transpose(1, 2), (pad, pad)).unsqueeze(2), (1, ks), stride=(1, stride))
    want = unf.view(B, F_ * ks, Tp).permute(2, 0, 1).reshape(Tp * B, F_ * ks)            # k = c * ks + kk
    assert torch.equal(col.cpu(), want)


@pytest.mark.parametrize('B,T,C,p', [(64, 79, 128, 0.2), (64, 80, 512, 0.0), (5, 43, 64, 0.2), (32, 8, 128, 0.0), (3, 30, 48, 0.2)])
def test_front_end_backward_in_two_launches(B, T, C, p):
    """halo_subsample_bwd with scratch lent (csrc/elementwise.hip, subsample_bwd_partial_kernel + subsample_bwd_reduce_kernel: mask,
    weight-gradient product and bias sums over row chunks, then their sum) against torch's float64 autograd of
    conv1d -> relu -> dropout (ha/rnn.py:16-19) with the mask read off the forward output, and against the four-launch path (no
    scratch lent): ragged chunks, rows % 4 != 0, fewer than 64 rows (falls back)."""
    from haloop_amd import _lib, ops
    _lib.lib()
    F_, ks, stride, pad = 80, 5, 4, 3
    g = torch.Generator().manual_seed(B * 3 + T)
    x = torch.randn(B, T, F_, generator=g)
    w = (torch.randn(C, F_, ks, generator=g) / 20).requires_grad_(True)
    bias = torch.randn(C, generator=g).requires_grad_(True)
    drop = ops.Dropout(p, 4321, 2, None) if p else ops.NO_DROPOUT
    y, col = ops.subsample_fwd(x.cuda(), w.detach().cuda(), bias.detach().cuda(), drop)
    Tp = y.shape[0]
    dy = torch.randn(Tp, B, C, generator=g)
    # reference: the same mask (y > 0 <=> relu passed and not dropped), float64
    pre = torch.nn.functional.conv1d(x.double().transpose(1, 2), w.double(), bias.double(), stride=stride, padding=pad).permute(2, 0, 1)
    mask = (y.cpu() > 0).double() * (1.0 / (1.0 - p) if p else 1.0)
    (pre * mask * dy.double()).sum().backward()
    res = {}
    for lend in (False, True):
        if lend:
            _lib.lend_scratch()
        else:
            _lib.check(_lib.lib().halo_set_scratch(None, 0), 'halo_set_scratch'); _lib._scratch = None
        dw, db = ops.subsample_bwd(dy.cuda(), y, col, B, T, F_, C, p)
        res[lend] = (dw.cpu(), db.cpu())
    for dw, db in res.values():
        np.testing.assert_allclose(dw.double().numpy(), w.grad.numpy(), rtol=0, atol=3e-6 * float(w.grad.abs().max()))
        np.testing.assert_allclose(db.double().numpy(), bias.grad.numpy(), rtol=0, atol=3e-6 * float(bias.grad.abs().max()))
    np.testing.assert_allclose(res[True][0].numpy(), res[False][0].numpy(), rtol=0, atol=2e-6 * float(w.grad.abs().max()))


def _normalised_close(a, b, rtol, atol):
    for k in a:
        scale = float(b[k].abs().max()) + 1e-12
        np.testing.assert_allclose(a[k].numpy() / scale, b[k].numpy() / scale, err_msg=k, rtol=rtol, atol=atol)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,p_drop,with_state', [
    (7, 5, 40, 256, 0.0, True),             # 16 workgroups, one ragged batch tile, carried state, exact-f32 input projection (in0 < 64)
    (6, 16, 128, 768, 0.25, False),         # inter-layer dropout: layer 1 reads the dropped image, the backward applies the mask
    (9, 33, 64, 512, 0.0, False),           # 3 batch tiles (plain block map), no dropout: layer 1 reads layer 0's own images
    (21, 64, 128, 1024, 0.2, False),        # the benchmark's grid: 256 workgroups, W_ih1 split between registers and LDS
    (1, 32, 128, 1024, 0.2, True),          # a single time step: combined steps 0 and 1 only
    (8, 128, 128, 1024, 0.2, False),        # 8 batch tiles on 256 CUs: two launches of 256 workgroups over the same buffers
    (5, 88, 64, 1024, 0.0, True),           # 6 tiles (the last ragged): launches of 4 and 2 tiles, carried state
    (4, 160, 128, 512, 0.1, False),         # H = 512: 8 tiles per launch, 10 tiles
    (251, 64, 128, 1024, 0.2, False),       # a 1000-frame utterance batch: 253 combined steps per launch, epochs and image offsets far from the benchmark's
])
def test_two_layer_launch_equals_layer_launches(hal, math_mode, T, B, in0, H, p_drop, with_state):
    """csrc/lstm_persist2.hip (both layers in one persistent launch per direction, bf16 mode) computes what the two per-layer
    persistent launches with the batched GEMMs between them compute: the same bf16 operands, fp32 sums in another order (a state
    element whose fp32 value moves across a bf16 rounding boundary moves its bf16 image by one unit: the bounds below)."""
    lib = hal['lib']
    assert lib.lib().halo_lstm_persistent2_eligible(T, B, H, 2) == 1
    a, st_a = _lstm_case(hal, T, B, in0, H, 2, p_drop, 5, with_state)
    lib.set_lstm_persistent2(False)
    try:
        assert lib.lib().halo_lstm_persistent2_eligible(T, B, H, 2) == 0
        b, st_b = _lstm_case(hal, T, B, in0, H, 2, p_drop, 5, with_state)
    finally:
        lib.set_lstm_persistent2(True)
    assert st_a == (0, 0) and st_b == (0, 0)
    _normalised_close(a, b, rtol=5e-3, atol=2e-3)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,p_drop,with_state', [
    (21, 128, 128, 1024, 0.2, False),       # the bench's B = 128: ONE launch of 256 workgroups, two tiles each (was two launches)
    (8, 256, 128, 1024, 0.2, False),        # 16 tiles: two interleaved launches of 8
    (5, 88, 64, 1024, 0.0, True),           # 6 tiles (the last ragged): 3 pairs = 192 workgroups, carried state
    (6, 104, 128, 1024, 0.2, False),        # 7 tiles (the last ragged): 6 interleaved, the seventh as a plain launch behind them
    (4, 144, 128, 1024, 0.1, True),         # 9 tiles: 8 interleaved, the ninth as a plain launch behind them
    (4, 160, 128, 512, 0.1, False),         # H = 512: 10 tiles, 5 pairs x 32 hidden tiles
    (3, 320, 64, 256, 0.0, False),          # H = 256: 20 tiles in one launch of 10 pairs x 16 hidden tiles (plain block map)
    (1, 128, 128, 1024, 0.2, True),         # a single time step
])
def test_interleaved_tiles_equal_consecutive_launches(hal, math_mode, T, B, in0, H, p_drop, with_state):
    """csrc/lstm_persist2x.hip (two batch tiles per workgroup, interleaved, when the batch has more tiles than one launch holds) against the
    consecutive launches of csrc/lstm_persist2.hip over the same buffers: the same products and sums in the same order per tile, so every
    output is BIT-identical -- but the bias gradients, whose per-tile rows the interleaved launch adds in registers (fp32 regrouping)."""
    lib = hal['lib']
    assert lib.lib().halo_lstm_persistent2_eligible(T, B, H, 2) == 1
    assert (B + 15) // 16 > 256 // (H // 16)                      # more tiles than a plain launch holds
    a, st_a = _lstm_case(hal, T, B, in0, H, 2, p_drop, 5, with_state)
    assert lib.lstm_chain_info('bwd')['kernel'] == 'lstm_persist2_bwd_kernel'
    lib.set_lstm_interleave(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, 2, p_drop, 5, with_state)
    finally:
        lib.set_lstm_interleave(True)
    assert st_a == (0, 0) and st_b == (0, 0)
    for k in a:
        if k.startswith('db_'):
            scale = float(b[k].abs().max()) + 1e-12
            np.testing.assert_allclose(a[k].numpy() / scale, b[k].numpy() / scale, rtol=0, atol=2e-6, err_msg=k)
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_interleaved_launches_see_fresh_data_through_reused_buffers(hal, math_mode):
    """The interleaved launches' hand-off images, epoch words and deferred publishes across back-to-back calls that re-use one reserve and one
    workspace (torch's allocator hands the same blocks back): alternating inputs, forward AND backward, every result bit for bit the one the
    consecutive launches give for THAT input -- never a piece left over from the call before (MI355X_MICROARCH.md: "test every hand-off ...
    consumer L1-warm, checking every word") -- and a 1000-frame batch (T' = 251: 253 epochs per tile, image offsets far from the benchmark's)."""
    lib, ops = hal['lib'], hal['ops']
    H, L, in0 = 1024, 2, 128
    g = torch.Generator().manual_seed(21)
    k = 1.0 / H ** 0.5
    w_ih = [((torch.rand(4 * H, in0 if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    drop = ops.Dropout(0.2, 99, 5)

    def run(x, dy):
        y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b, drop=drop)
        ws = ops.lstm_bwd_workspace(x, w_hh)
        dx, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (x.shape[1] * H, H), False, reserve, drop=drop, workspace=ws, want_dx=True)
        return [y.clone(), dx.clone(), grads['dw_hh'][0].clone(), grads['dw_ih'][1].clone()]

    for T, B, reps in ((21, 128, 3), (251, 128, 1)):
        xs = [torch.randn(T, B, in0, generator=g).to(DEV) * s for s in (1.0, -0.7, 1.9)]
        dys = [torch.randn(T, B, H, generator=g).to(DEV) * s for s in (0.5, 1.0, -0.3)]
        assert lib.lib().halo_lstm_persistent2_eligible(T, B, H, L) == 1
        lib.set_lstm_interleave(False)
        try:
            refs = [run(x, dy) for x, dy in zip(xs, dys)]
        finally:
            lib.set_lstm_interleave(True)
        for rep in range(reps):
            for i, (x, dy) in enumerate(zip(xs, dys)):
                got = run(x, dy)
                for a, r in zip(got, refs[i]):
                    assert torch.equal(a, r), (T, rep, i)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_a_mute_workgroup_times_the_interleaved_launch_out_visibly(hal, math_mode):
    """The interleaved forward with one workgroup that never publishes (halo_debug_mute_workgroup): its peers' bounded waits time out, the
    launch drains (no hang: every wave leaves behind barrier AC), the call's abort word and the caller's sticky status word are raised; the
    next call, unmuted, is clean and gives the consecutive launches' bits."""
    lib, ops = hal['lib'], hal['ops']
    T, B, in0, H, L = 3, 128, 128, 1024, 2
    g = torch.Generator().manual_seed(31)
    k = 1.0 / H ** 0.5
    x = torch.randn(T, B, in0, generator=g).to(DEV)
    w_ih = [((torch.rand(4 * H, in0 if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b = [torch.zeros(4 * H, device=DEV) for l in range(L)]
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    lib.set_status_word(status)
    try:
        lib.check(lib.lib().halo_debug_mute_workgroup(5), 'mute')
        y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b)
        torch.cuda.synchronize()
        assert _status(hal, reserve, False, T, B, in0, H, L) != 0 and int(status.item()) != 0
        lib.check(lib.lib().halo_debug_mute_workgroup(-1), 'unmute')
        status.zero_()
        y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b)
        assert _status(hal, reserve, False, T, B, in0, H, L) == 0 and int(status.item()) == 0
        lib.set_lstm_interleave(False)
        try:
            y2 = ops.lstm_fwd(x, w_ih, w_hh, b, b)[0]
        finally:
            lib.set_lstm_interleave(True)
        assert torch.equal(y, y2)
    finally:
        lib.lib().halo_debug_mute_workgroup(-1)
        lib.set_status_word(None)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_split_backward_follows_the_two_layer_forward(hal, math_mode):
    """The two-layer forward leaves the reserve the per-layer backward expects: backward called layer by layer (the data-parallel
    step does, haloop_amd/train.py) after the fused forward equals the one-call two-layer backward."""
    ops, lib = hal['ops'], hal['lib']
    T, B, in0, H, L = 8, 32, 128, 512, 2
    g = torch.Generator().manual_seed(11)
    k = 1.0 / H ** 0.5
    x = torch.randn(T, B, in0, generator=g).to(DEV)
    w_ih = [((torch.rand(4 * H, in0 if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_ih = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_hh = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    dy = torch.randn(T, B, H, generator=g).to(DEV)
    drop = ops.Dropout(0.2, 77, 1)
    outs = []
    for split in (False, True):
        y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh, drop=drop)
        ws = ops.lstm_bwd_workspace(x, w_hh)
        if split:
            _, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, drop=drop, workspace=ws, layers=(1, 2))
            dx, grads = ops.lstm_bwd(x, w_ih, w_hh, None, (B * H, H), False, reserve, drop=drop, workspace=ws, layers=(0, 1), grads=grads,
                                     want_dx=True)
        else:
            dx, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, drop=drop, workspace=ws, want_dx=True)
        out = {'y': y.cpu(), 'dx': dx.cpu()}
        for name, lst in grads.items():
            for l, t in enumerate(lst):
                out[f'{name}{l}'] = t.cpu()
        outs.append(out)
    _normalised_close(outs[0], outs[1], rtol=5e-3, atol=2e-3)


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
@pytest.mark.parametrize('T,B,C,H', [(20, 64, 128, 1024), (7, 32, 128, 256)])
def test_input_gradient_left_as_k_slices_for_the_conv_backward(hal, math_mode, T, B, C, H):
    """ops.lstm_bwd(dx_slabs=n) leaves the two-layer launch's input gradient as unreduced K-slices (no split-K reduce launch) and
    ops.subsample_bwd(slabs=) adds them while reading: the slices sum to the plain call's dx (another summation order of the same
    products), and the conv's weight / bias gradients from the slices equal those from the summed matrix."""
    ops, lib = hal['ops'], hal['lib']
    L, F_, ks = 2, 16, 5
    g = torch.Generator().manual_seed(5)
    k = 1.0 / H ** 0.5
    x = torch.relu(torch.randn(T, B, C, generator=g)).to(DEV)            # the conv's relu output: its sign pattern is the mask
    col = torch.randn(T * B, F_ * ks, generator=g).to(DEV)
    w_ih = [((torch.rand(4 * H, C if l == 0 else H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    w_hh = [((torch.rand(4 * H, H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_ih = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    b_hh = [((torch.rand(4 * H, generator=g) * 2 - 1) * k).to(DEV) for l in range(L)]
    dy = torch.randn(T, B, H, generator=g).to(DEV)
    _, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh)
    ws = ops.lstm_bwd_workspace(x, w_hh)
    dx_plain, grads_plain = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, workspace=ws, want_dx=True)
    assert ops.lstm_dx_slabs_left() == 1
    _, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh)
    n = 16
    slabs = torch.full((n, T, B, C), float('nan'), device=DEV)
    _, grads = ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, reserve, workspace=ws, dx=slabs, dx_slabs=n)
    left = ops.lstm_dx_slabs_left()
    two_layer = bool(lib.lib().halo_lstm_persistent2_eligible(T, B, H, L)) and math_mode == 'bf16'
    assert (left > 1) == two_layer and left <= n
    summed = slabs[:left].sum(0)
    assert torch.isfinite(summed).all()
    scale = dx_plain.abs().max().item()
    assert (summed - dx_plain).abs().max().item() <= 2e-5 * scale + 1e-7
    for name in grads:
        for a, b in zip(grads[name], grads_plain[name]):
            assert torch.equal(a, b)
    Tin = 4 * (T - 1) + ks - 6          # a frame count whose subsampled length is T
    dw_a, db_a = ops.subsample_bwd(slabs, x, col, B, Tin, F_, C, 0.0, slabs=left)
    dw_b, db_b = ops.subsample_bwd(summed.contiguous(), x, col, B, Tin, F_, C, 0.0)
    assert torch.allclose(dw_a, dw_b, rtol=1e-5, atol=1e-5 * dw_b.abs().max().item())
    assert torch.allclose(db_a, db_b, rtol=1e-5, atol=1e-5 * db_b.abs().max().item())
    assert ops.lstm_bwd(x, w_ih, w_hh, dy, (B * H, H), False, ops.lstm_fwd(x, w_ih, w_hh, b_ih, b_hh)[3], workspace=ws, want_dx=True)[0] is not None
    assert ops.lstm_dx_slabs_left() == 1                # the setting does not outlive the call that asked for it


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
def test_two_train_mode_steps_at_config2_match_the_oracle_with_the_same_masks(hal, math_mode):
    """The benchmarked configuration itself -- LC-2x1024, B=64, dropout 0.2 in all three places (ha/rnn.py:8,11, ha/recognizer.py:41),
    the whole step replayed from one HIP graph with the device-side dropout counter, the persistent recurrences (bf16: the two-layer
    launch) -- against the CPU restatement's Trainer fed the SAME Philox masks (counter values 0 and 1): losses, clipped gradient
    norms, and the weights after the two AdamW updates.  bf16x3: the fp32-grade bounds; bf16: SURVEY.md 8d's bf16-MFMA bound on the
    loss (2e-2), and weights that moved by at most what a flipped sign of a vanishing gradient moves them (2 lr per step)."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    F_, C, H, L, V, B, T, S = 80, 128, 1024, 2, 32, 64, 80, 10
    lr, seed = 3e-4, 20240607
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 42)
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).train(); rec.to(DEV).train()
    tr = LstmCtcTrainer(enc, rec, lr=lr, seed=seed, use_graph=True)
    ref = cpu_ref.Trainer(enc_p, rec_p, lr=lr)
    Tp = (T + 6 - 5) // 4 + 1
    exact = math_mode == 'bf16x3'
    for step in range(2):
        x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 300 + step)
        masks = cpu_ref.philox_masks(B, Tp, C, H, L, 0.2, 0.2, seed, step)
        loss_ref, gnorm_ref = ref.step(x, il, tg, tl, masks=masks)
        loss = tr.step(x.to(DEV), il.to(DEV), tg.to(DEV), tl.to(DEV))
        np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=2e-5 if exact else 2e-2)
        np.testing.assert_allclose(tr.grad_norm.item(), gnorm_ref.item(), rtol=1e-4 if exact else 5e-2)
    assert int(tr.counter.item()) == 2 and int(tr.adam_step.item()) == 2
    sd = {**{'encoder.' + k: v for k, v in enc.state_dict().items()}, **{'recognizer.' + k: v for k, v in rec.state_dict().items()}}
    want = {**{'encoder.' + k: v for k, v in ref.enc.items()}, **{'recognizer.' + k: v for k, v in ref.rec.items()}}
    for k, v in sd.items():
        a, b = v.reshape(-1)[::997].cpu().numpy(), want[k].detach().reshape(-1)[::997].numpy()
        if exact:
            # (Adam's normalised update makes an element whose gradient is within rounding of zero move by up to lr either way:
            # nearly all elements within 2e-5, none further than 1e-4)
            assert (np.abs(a - b) <= 2e-5).mean() >= 0.999, (k, (np.abs(a - b) <= 2e-5).mean())
            np.testing.assert_allclose(a, b, atol=1e-4, err_msg=k)
        else:
            d = np.abs(a - b)
            assert d.max() <= 4 * lr * 1.01 + 1e-6, (k, d.max())
            assert (d <= 3e-5).mean() >= 0.97, (k, (d <= 3e-5).mean())


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
@pytest.mark.parametrize('use_graph', [True, False])
def test_a_timed_out_recurrence_is_visible_and_applies_no_update(hal, math_mode, use_graph):
    """A persistent recurrence whose bounded wait times out (here: one workgroup made mute by the test hook, so its peers never see
    its epoch) raises the caller's sticky status word; the clip kernel then treats the step like a non-finite one -- no update, the
    Adam step count stays -- and LstmCtcTrainer.check_status() raises.  After the word is cleared training continues."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    lib = hal['lib']
    F_, C, H, L, V, B, T, S = 40, 64, 256, 2, 16, 16, 40, 4
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 3)
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).train(); rec.to(DEV).train()
    tr = LstmCtcTrainer(enc, rec, lr=1e-3, seed=5, use_graph=use_graph)
    x, il, tg, tl = (t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 8))
    assert lib.lib().halo_lstm_persistent_eligible(B, H) == 1
    try:
        tr.step(x, il, tg, tl)
        tr.check_status()
        assert int(tr.adam_step.item()) == 1
        before = tr.flat.params.clone()
        lib.check(lib.lib().halo_debug_mute_workgroup(3), 'mute')
        if use_graph:
            tr._graphs = None                      # the hook is a launch argument: record the step again with it
        tr.step(x, il, tg, tl)
        assert int(tr.status.item()) != 0
        assert int(tr.adam_step.item()) == 1 and torch.equal(tr.flat.params, before)
        with pytest.raises(lib.HaloError):
            tr.check_status()
        lib.check(lib.lib().halo_debug_mute_workgroup(-1), 'unmute')
        if use_graph:
            tr._graphs = None
        tr.step(x, il, tg, tl)                     # the word is sticky: still no update
        assert int(tr.adam_step.item()) == 1 and torch.equal(tr.flat.params, before)
        tr.status.zero_()
        tr.step(x, il, tg, tl)
        tr.check_status()
        assert int(tr.adam_step.item()) == 2 and not torch.equal(tr.flat.params, before)
    finally:
        lib.lib().halo_debug_mute_workgroup(-1)
        lib.set_status_word(None)


def test_learning_rate_changes_reach_the_captured_step(hal):
    """The optimizer launch inside the step graph reads the learning rate from the device: assigning trainer.lr between steps (the
    reference applies its schedule every step, ha/loop.py:191) changes the replayed update exactly as it changes the eager one."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    F_, C, H, L, V, B, T, S = 40, 64, 64, 2, 16, 8, 40, 4
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 4)
    finals = []
    for use_graph in (True, False):
        enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        enc.to(DEV).eval(); rec.to(DEV).eval()
        tr = LstmCtcTrainer(enc, rec, lr=1e-3, use_graph=use_graph)
        for step, lr in enumerate((1e-3, 5e-4, 0.0)):
            tr.lr = lr
            x, il, tg, tl = (t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 30 + step))
            before = tr.flat.params.clone()
            tr.step(x, il, tg, tl)
            if lr == 0.0:
                assert torch.equal(tr.flat.params, before)       # lr = 0: decay factor 1, step size 0
        finals.append(tr.flat.params.clone())
    hal['lib'].set_status_word(None)
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,L,p_drop,with_state', [
    (21, 128, 128, 1024, 2, 0.2, False),    # the bench's b_sweep point: 64 hidden tiles x 4 thirty-two-row tiles = 256 workgroups
    (6, 80, 64, 1024, 1, 0.0, True),        # five 16-row tiles: the last workgroup row has ONE sub-tile; carried state
    (4, 100, 96, 768, 2, 0.25, False),      # ragged last sub-tile (100 = 6 * 16 + 4), 48 x 4 workgroups, dropout between the layers
    (5, 64, 128, 1536, 1, 0.0, False),      # the reference's H = 1536 variant (ha/init.py:171): 96 hidden tiles x 2 thirty-two-row tiles
])
def test_wide_persistent_recurrence_equals_step_chain(hal, math_mode, T, B, in0, H, L, p_drop, with_state):
    """csrc/lstm_persist32.hip (32 batch rows per workgroup, bf16 arithmetic: batches the 16-row grid cannot hold with one workgroup
    per CU) against the step-launch chain: same function, another summation order."""
    lib = hal['lib']
    cus = 256
    assert (H // 16) * ((B + 15) // 16) > cus >= (H // 16) * ((B + 31) // 32)          # the 16-row grid does not fit, the 32-row grid does
    assert lib.lib().halo_lstm_persistent_eligible(B, H) == 1
    a, st_a = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    assert 'persist' in lib.lstm_chain_info('bwd')['kernel']
    lib.set_lstm_persistent(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
        assert 'step' in lib.lstm_chain_info('bwd')['kernel']
    finally:
        lib.set_lstm_persistent(True)
    assert st_a == (0, 0) and st_b == (0, 0)
    _normalised_close(a, b, rtol=3e-2, atol=3e-3)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,L,p_drop,with_state', [
    (6, 16, 64, 256, 3, 0.2, False),        # the stock 3-layer encoder's shape family (ha/rnn.py:11): layer 0 alone, layers 1 + 2 in one launch
    (5, 32, 128, 512, 3, 0.0, True),        # carried state, no dropout
    (4, 32, 128, 256, 4, 0.25, False),      # two single layers under the pair
    (21, 64, 128, 1024, 3, 0.2, False),     # the stock encoder itself at BASELINE config 2's grid (bench.py's `stock3` leg)
])
def test_top_pair_of_a_deeper_stack_runs_as_one_launch(hal, math_mode, T, B, in0, H, L, p_drop, with_state):
    """L > 2 in bf16: the stack's top two layers run as the two-layer persistent launch (its lower layer's input projection fed by the layer
    below, its input gradient masked by that layer's dropout and handed down), the layers below one launch each -- against the per-layer
    launches and against the step chain."""
    lib = hal['lib']
    assert lib.lib().halo_lstm_persistent2_eligible(T, B, H, L) == 1
    a, st_a = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    assert lib.lstm_chain_info('fwd')['kernel'] == 'lstm_persist2_fwd_kernel'
    lib.set_lstm_persistent2(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    finally:
        lib.set_lstm_persistent2(True)
    lib.set_lstm_persistent(False)
    try:
        c, st_c = _lstm_case(hal, T, B, in0, H, L, p_drop, 5, with_state)
    finally:
        lib.set_lstm_persistent(True)
    assert st_a == (0, 0) and st_b == (0, 0) and st_c == (0, 0)
    _normalised_close(a, b, rtol=5e-3, atol=2e-3)
    _normalised_close(a, c, rtol=3e-2, atol=3e-3)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('T,B,in0,H,L,p_drop,with_state', [
    (21, 64, 128, 1024, 2, 0.2, False),     # the benchmark's grid: 128 + 80 tiles of 256 x 256
    (6, 16, 128, 768, 2, 0.25, False),      # H = 768: the outputs are cut at column 768 = 3 tiles; the lower layer's last tile is half empty
    (7, 5, 40, 256, 2, 0.0, True),          # 40 input features: the second output is 40 columns wide, its row block the image's last
    (5, 32, 128, 512, 3, 0.0, True),        # the top pair of a deeper stack
    (43, 128, 128, 1024, 2, 0.2, False),    # K = 5504: the interleaved launches' images
    (9, 32, 128, 1536, 2, 0.2, False),      # H = 1536: one layer per launch; 288 and 168 tiles of 256 x 256 (the second round of the first: 32)
])
def test_weight_gradients_on_256_tiles_equal_the_128_tile_launches(hal, math_mode, T, B, in0, H, L, p_drop, with_state):
    """csrc/gemm256.h (both layers' dW_hh | dW_ih in ONE launch of 256 x 256 tiles) against the two 128 x 128-tile launches it replaces: the
    same operand images, the same bf16 products, fp32 sums in a different order."""
    lib = hal['lib']
    a, st_a = _lstm_case(hal, T, B, in0, H, L, p_drop, 9, with_state)
    lib.set_gemm256(False)
    try:
        b, st_b = _lstm_case(hal, T, B, in0, H, L, p_drop, 9, with_state)
    finally:
        lib.set_gemm256(True)
    assert st_a == (0, 0) and st_b == (0, 0)
    for k in a:
        if k.startswith('dw_'):
            scale = float(b[k].abs().max()) + 1e-12
            np.testing.assert_allclose(a[k].numpy() / scale, b[k].numpy() / scale, rtol=0, atol=2e-6, err_msg=k)
        else:
            assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_training_step_with_the_input_gradient_slices_in_the_256_tile_launch(hal, math_mode):
    """The trainer's step hands the conv backward K-slices of the LSTM's input gradient (halo_set_lstm_dx_slabs): with gemm256 they ride
    as the third problem of the one launch.  Two steps from the same weights, with and without it, agree to summation order."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    F_, C, H, L, V, B, T, S = 80, 128, 1024, 2, 32, 64, 80, 10
    res = []
    for on in (True, False):
        hal['lib'].set_gemm256(on)
        try:
            enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 7)
            enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
            enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
            enc.to(DEV).train(); rec.to(DEV).train()
            tr = LstmCtcTrainer(enc, rec, lr=1e-3, use_graph=False, seed=5)
            for step in range(2):
                batch = tuple(t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 300 + step))
                loss = tr.step(*batch)
            tr.check_status()
            hal['lib'].set_status_word(None)
            res.append((loss.item(), tr.grad_norm.item(), {k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}))
        finally:
            hal['lib'].set_gemm256(True)
    np.testing.assert_allclose(res[0][0], res[1][0], rtol=1e-5)
    np.testing.assert_allclose(res[0][1], res[1][1], rtol=1e-4)
    for k in res[0][2]:
        np.testing.assert_allclose(res[0][2][k].numpy(), res[1][2][k].numpy(), atol=2e-5, err_msg=k)


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
@pytest.mark.parametrize('use_graph', [True, False])
def test_recognizer_keeps_packed_weights_only_while_the_weights_stand(hal, math_mode, use_graph):
    """infer.LstmCtcRecognizer at the two-layer launch's shape: the packed weight images stay in the recognizer's reserve between calls
    (halo_set_lstm_weights_stamp) -- and are rebuilt after training steps, whose optimizer launches write the parameters behind torch's
    version counters, and after an in-place torch update: every result equals that of a recognizer built fresh on the same weights."""
    from oracle import cpu_ref
    from haloop_amd.infer import LstmCtcRecognizer
    from haloop_amd.train import LstmCtcTrainer
    F_, C, H, L, V, B, T, S = 40, 64, 256, 2, 16, 32, 40, 4
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 4)
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    assert hal['lib'].lib().halo_lstm_persistent2_eligible(10, B, H, L) == 1
    x = cpu_ref.synthetic_batch(B, T, F_, V, S, 77)[0].to(DEV)
    reco = LstmCtcRecognizer(enc, rec, use_graph=use_graph)

    def same_as_fresh():
        got = [o.clone() for o in reco.recognize(x)]
        again = [o.clone() for o in reco.recognize(x)]                      # (second call: the images are kept)
        want = LstmCtcRecognizer(enc, rec, use_graph=False).recognize(x)
        for a, b, c in zip(got, again, want):
            assert torch.equal(a, c) and torch.equal(b, c)
        return got

    first = same_as_fresh()
    tr = LstmCtcTrainer(enc, rec, lr=3e-2, use_graph=True)
    for step in range(3):                                                    # (the third replays the captured step)
        xb, il, tg, tl = (t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 30 + step))
        tr.step(xb, il, tg, tl)
    hal['lib'].set_status_word(None)
    after = same_as_fresh()
    assert not torch.equal(first[1], after[1])                               # the scores moved with the weights
    with torch.no_grad():
        enc.lstm.weight_hh_l1.mul_(0.5)
    moved = same_as_fresh()
    assert not torch.equal(after[1], moved[1])


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_trainer_auto_launch_mode_settles_and_keeps_the_trajectory(hal, math_mode):
    """LstmCtcTrainer(use_graph='auto') times graph replays and eager launches over its first 53 steps and keeps the faster way: the
    losses are those of the graph-mode trainer step for step (the same launches either way), and a choice has been recorded."""
    from haloop_amd.train import LstmCtcTrainer
    from haloop_amd import synth
    F_, C, H, L, V, B, T, S = 80, 128, 512, 2, 32, 32, 80, 8
    x, il, tg, tl = [t.to(DEV) for t in synth.synthetic_batch(B, T, F_, V, S, 3)]
    losses = {}
    for mode in (True, 'auto'):
        enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
        enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        enc.to(DEV).train(); rec.to(DEV).train()
        tr = LstmCtcTrainer(enc, rec, lr=3e-4, seed=5, use_graph=mode)
        losses[mode] = torch.stack([tr.step(x, il, tg, tl) for _ in range(60)]).cpu()
        tr.check_status()
        if mode == 'auto':
            assert tr._auto is None and tr.auto_choice is not None and isinstance(tr.use_graph, bool)
            assert tr.auto_choice['graph_replay_ms'] > 0 and tr.auto_choice['eager_launches_ms'] > 0
    assert torch.equal(losses[True], losses['auto'])


@pytest.mark.parametrize('math_mode', ['bf16'], indirect=True)
def test_recognizer_auto_launch_mode(hal, math_mode):
    """LstmCtcRecognizer(use_graph='auto'): 46 calls settle the launch mode; every call's result equals the graph-mode recognizer's."""
    from haloop_amd.infer import LstmCtcRecognizer
    from haloop_amd import synth
    F_, C, H, L, V, B, T = 80, 128, 512, 2, 32, 32, 80
    enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV); rec.to(DEV)
    ref = LstmCtcRecognizer(enc, rec, use_graph=True)
    auto = LstmCtcRecognizer(enc, rec, use_graph='auto')
    for i in range(50):
        x = synth.synthetic_batch(B, T, F_, V, 8, 100 + i % 3)[0].to(DEV)
        want, got = ref.recognize(x), auto.recognize(x)
        for a, b in zip(want, got):
            assert torch.equal(a, b), i
    assert auto._auto is None and auto.auto_choice is not None and isinstance(auto.use_graph, bool)


@pytest.mark.parametrize('math_mode', ['bf16', 'bf16x3'], indirect=True)
def test_trainer_on_long_utterances(hal, math_mode):
    """400-frame utterances (T' = 101: the head runs as separate operators, the LSTM's small sums still ride in the conv backward's launch):
    three trainer steps replayed from the graph and launched eagerly give the same losses bit for bit, finite and decreasing on a repeated batch."""
    from haloop_amd.train import LstmCtcTrainer
    from haloop_amd import synth
    F_, C, H, L, V, B, T, S = 40, 64, 256, 2, 32, 16, 400, 12
    x, il, tg, tl = [t.to(DEV) for t in synth.synthetic_batch(B, T, F_, V, S, 4)]
    out = {}
    for graph in (True, False):
        enc_p, rec_p = synth.make_params(F_, C, H, L, V, 42)
        enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        enc.to(DEV).train(); rec.to(DEV).train()
        tr = LstmCtcTrainer(enc, rec, lr=1e-3, seed=9, use_graph=graph)
        out[graph] = torch.stack([tr.step(x, il, tg, tl) for _ in range(6)]).cpu()
        tr.check_status()
    assert torch.equal(out[True], out[False])
    assert torch.isfinite(out[True]).all() and out[True][-1] < out[True][0]
