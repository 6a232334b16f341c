"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference-generated
golden fixtures.  Integer outputs must be exact; floating point within the stated tolerances:

    fp32-grade modes ('f32' = exact-f32 MFMA, 'bf16x3' = split-bf16, both run by the BOTH_MODES tests): loss rel <= 1e-5,
    features max-abs <= 1e-4, logits-gradient max-abs <= 1e-5 (the reference's own two CTC
    implementations agree to 5-8e-6).  'bf16' mode (operands rounded to bf16): the bf16-MFMA tolerance of the same
    section (loss rel <= 2e-2), in the tests that name it.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

DEV = 'cuda'


@pytest.fixture(scope='module')
def hal():
    import haloop_amd
    from haloop_amd import _lib, ops, functional, rnn, recognizer, ctc, beam
    _lib.lib()           # raises if libhalo.so is missing: there is no fallback to test
    _lib.lend_scratch()  # exercise the split-K path of the under-filled GEMMs
    return dict(ops=ops, F=functional, rnn=rnn, recognizer=recognizer, ctc=ctc, beam=beam, lib=_lib)


def _unpad(seqs, lens):
    return [list(map(int, s[:n])) for s, n in zip(seqs, lens)]


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize('akc,bkc', [(1, 1), (1, 0), (0, 0), (0, 1)])
@pytest.mark.parametrize('M,N,K', [(1344, 4096, 128), (256, 200, 400), (70, 32, 1024), (4096, 1024, 84), (33, 17, 5)])
def test_gemm_layouts(hal, akc, bkc, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    ref = (a.double() @ b.double() + bias.double()).clamp_min(0)
    A = (a if akc else a.t().contiguous()).to(DEV)
    B = (b.t().contiguous() if bkc else b).to(DEV)
    out = hal['ops'].gemm(A, B, akc, bkc, M, N, K, bias1=bias.to(DEV), relu=True)
    err = (out.cpu().double() - ref).abs().max().item()
    assert err <= 2e-6 * K ** 0.5 * 8, err


# -------------------------------------------------------------------------------------------- CTC
@pytest.mark.parametrize('case', ['random', 'repeat', 's1', 'ragged', 'infeasible', 'wide'])
def test_ctc_against_reference_goldens(hal, case):
    g = load_golden('g2_ctc')
    logits = torch.from_numpy(g[case + '.logits']).to(DEV).requires_grad_(True)
    tg, il, tl = (torch.from_numpy(g[case + '.' + k]).to(DEV) for k in ('targets', 'il', 'tl'))
    em = hal['F'].log_softmax(logits)
    # ha.ctc.ctc_forward_score3 surface
    s3 = hal['ctc'].ctc_forward_score3(em.detach(), tg, il, tl).cpu().numpy()
    np.testing.assert_allclose(s3, g[case + '.score3'], rtol=1e-5)
    np.testing.assert_allclose(hal['ctc'].ctc_reduce_mean(torch.from_numpy(s3), tl.cpu()).numpy(),
                               g[case + '.reduce_mean3'], rtol=1e-5)
    # F.ctc_loss surface
    nll = hal['F'].ctc_loss(em, tg, il, tl, reduction='none')
    ref = g[case + '.torch_none']
    ok = np.isfinite(ref)
    np.testing.assert_allclose(nll.detach().cpu().numpy()[ok], ref[ok], rtol=1e-5)
    assert np.array_equal(np.isinf(nll.detach().cpu().numpy()), ~ok)
    if bool(g[case + '.has_grad']):
        loss = hal['F'].ctc_loss(em, tg, il, tl, reduction='mean')
        np.testing.assert_allclose(loss.item(), float(g[case + '.torch_mean']), rtol=1e-5)
        loss.backward()
        np.testing.assert_allclose(logits.grad.cpu().numpy(), g[case + '.dlogits_mean'], atol=1e-5)


def test_ctc_single_sequence_variants(hal):
    g = load_golden('g2_ctc')
    for pre in ('demo', 'wrap'):
        l = torch.from_numpy(g[pre + ('.l0' if pre == 'demo' else '.l')]).to(DEV)
        t = torch.from_numpy(g[pre + ('.t0' if pre == 'demo' else '.t')]).to(DEV)
        np.testing.assert_allclose(hal['ctc'].ctc_forward_score1(l, t).item(), float(g[pre + '.score1']), rtol=1e-5)
        np.testing.assert_allclose(hal['ctc'].ctc_forward_score2(l, t).item(), float(g[pre + '.score2']), rtol=1e-5)


def test_ctc_lattice_vs_oracle_random_sizes(hal):
    from oracle import lattice
    gen = torch.Generator().manual_seed(11)
    for T, N, C, S in [(1, 1, 3, 1), (7, 5, 4, 3), (64, 3, 11, 31), (130, 2, 40, 70), (21, 64, 32, 10)]:
        em = torch.randn(T, N, C, generator=gen).log_softmax(-1)
        tg = torch.randint(1, C, (N, S), generator=gen)
        il = torch.randint(max(1, T // 2), T + 1, (N,), generator=gen)
        tl = torch.randint(1, S + 1, (N,), generator=gen)
        want = lattice.ctc_forward_score3(em, tg, il, tl).numpy()
        got = hal['ctc'].ctc_forward_score3(em.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV)).cpu().numpy()
        np.testing.assert_allclose(got, want, rtol=2e-5)
        want_t = torch.nn.functional.ctc_loss(em, tg, il, tl, reduction='none').numpy()
        got_t = hal['F'].ctc_loss(em.to(DEV), tg.to(DEV), il.to(DEV), tl.to(DEV), reduction='none').cpu().numpy()
        fin = np.isfinite(want_t)
        np.testing.assert_allclose(got_t[fin], want_t[fin], rtol=2e-5)
        assert np.array_equal(np.isinf(got_t), ~fin)


def test_ctc_gradient_vs_torch_cpu(hal):
    gen = torch.Generator().manual_seed(5)
    T, N, C, S = 33, 6, 9, 8
    logits = torch.randn(T, N, C, generator=gen)
    tg = torch.randint(1, C, (N, S), generator=gen)
    il = torch.tensor([33, 30, 20, 33, 17, 25])
    tl = torch.tensor([8, 3, 5, 1, 7, 2])
    lc = logits.clone().requires_grad_(True)
    torch.nn.functional.ctc_loss(lc.log_softmax(-1), tg, il, tl).backward()
    lg = logits.to(DEV).requires_grad_(True)
    hal['F'].ctc_loss(hal['F'].log_softmax(lg), tg.to(DEV), il.to(DEV), tl.to(DEV)).backward()
    np.testing.assert_allclose(lg.grad.cpu().numpy(), lc.grad.numpy(), atol=1e-5)   # stated logits-grad tolerance


def test_ctc_edge_cases_vs_torch_cpu(hal):
    """The lattice's corners against stock torch CPU ``F.ctc_loss`` (what the reference runs, ha/recognizer.py:71): an EMPTY target (only
    blanks may be emitted), a single frame, exactly as many frames as labels, repeated labels that need separating blanks (one frame
    short: infeasible, inf and a zero gradient row... torch gives nan there: compared as 'not finite'), the longest target the batch
    holds, and a batch of one -- per-utterance losses (reduction='none') and the gradient at the logits of the feasible utterances."""
    gen = torch.Generator().manual_seed(12)
    T, C, S = 12, 7, 6
    tg = torch.tensor([[1, 2, 3, 4, 5, 6],      # tl = 0: empty target
                       [3, 0, 0, 0, 0, 0],      # one label, one frame
                       [1, 2, 3, 4, 0, 0],      # il == tl: every frame a label
                       [2, 2, 2, 0, 0, 0],      # repeats: needs 2 tl - 1 = 5 frames, has 5
                       [2, 2, 2, 0, 0, 0],      # ... has 4: infeasible
                       [6, 5, 6, 5, 6, 5]])     # the full width S, all frames
    il = torch.tensor([7, 1, 4, 5, 4, 12])
    tl = torch.tensor([0, 1, 4, 3, 3, 6])
    for N in (6, 1):
        logits = torch.randn(T, N, C, generator=gen)
        lc = logits.clone().requires_grad_(True)
        want = torch.nn.functional.ctc_loss(lc.log_softmax(-1), tg[:N], il[:N], tl[:N], reduction='none')
        fin = torch.isfinite(want)
        want[fin].sum().backward()
        lg = logits.to(DEV).requires_grad_(True)
        got = hal['F'].ctc_loss(hal['F'].log_softmax(lg), tg[:N].to(DEV), il[:N].to(DEV), tl[:N].to(DEV), reduction='none')
        assert torch.equal(torch.isfinite(got).cpu(), fin), (got, want)
        np.testing.assert_allclose(got.detach().cpu().numpy()[fin.numpy()], want.detach().numpy()[fin.numpy()], rtol=2e-6, atol=2e-6)
        got[fin.to(DEV)].sum().backward()
        np.testing.assert_allclose(lg.grad.cpu().numpy()[:, fin.numpy()], lc.grad.numpy()[:, fin.numpy()], atol=1e-5)
        if N == 6:
            assert not fin[4] and fin[[0, 1, 2, 3, 5]].all()
            assert abs(want[0].item() + lc.log_softmax(-1)[:7, 0, 0].sum().item()) < 1e-5      # empty target: -sum of the blank log-probs


# ----------------------------------------------------------------------------------- greedy / beam
def test_greedy_matches_reference(hal):
    g = load_golden('g1_tiny_l2')
    lp = torch.from_numpy(g['lp']).to(DEV)
    ali, scores, hyp, hlen = hal['ops'].ctc_greedy(lp)
    assert np.array_equal(ali.cpu().numpy(), g['ali'])
    np.testing.assert_array_equal(scores.cpu().numpy(), g['scores'])
    assert np.array_equal(hlen.cpu().numpy(), g['hlen'])
    got = [hyp[i, :n].tolist() for i, n in enumerate(hlen.tolist())]
    assert got == _unpad(g['hyps'], g['hlen'])


def test_greedy_vs_oracle_long(hal):
    from oracle import lattice
    gen = torch.Generator().manual_seed(2)
    lp = (torch.randn(5, 150, 4, generator=gen) * 3).log_softmax(-1)      # T > 64: multi-pass carry
    hyps, lens, ali, _ = lattice.greedy_decode(lp)
    a, s, h, hl = hal['ops'].ctc_greedy(lp.to(DEV))
    assert np.array_equal(a.cpu().numpy(), ali.numpy())
    assert hl.tolist() == lens.tolist()
    assert [h[i, :n].tolist() for i, n in enumerate(hl.tolist())] == hyps


# Beam token ids are required EXACT on every reference-generated case.  On flat random emissions hypotheses that differ only in
# early symbols converge to bit-identical fp32 scores, and which of them the reference keeps depends on the last ulp of
# torch.logaddexp in the frames before they meet; the kernel therefore evaluates logaddexp the way ATen's CPU kernel does (Sleef's
# expf / log1pf for whole vector chunks of the candidate array, glibc's for the remainder and the 0-dim calls), restated bit for
# bit in csrc/beam.hip.  The fixtures were generated on an AVX-512 machine (chunk 32, the library's default).
ALL_BEAM = ['r6x4b4', 'r21x32b3', 'p40x32b8', 'p21x256b4', 'onehot', 'r21x32b16', 'r30x9b5', 'r21x32b33', 'p21x32b16', 'p64x9b9']


@pytest.mark.parametrize('case', ALL_BEAM)
def test_beam_logits_matches_reference(hal, case):
    g = load_golden('g3_beam')
    hal['beam'].set_reference_cpu('AVX512')
    seqs, scores = hal['beam'].ctc_beam_search_decode_logits(torch.from_numpy(g[case + '.logits']).to(DEV),
                                                             int(g[case + '.beam']))
    assert seqs == _unpad(g[case + '.seqs'], g[case + '.lens'])            # token ids: exact
    np.testing.assert_array_equal(scores.cpu().numpy(), g[case + '.scores'])   # and so are the scores, bit for bit


def test_logaddexp_replica_matches_torch_cpu(hal):
    """halo_logaddexp_aten == torch.logaddexp on this host's CPU, bit for bit: vector chunks (Sleef) and scalar remainder (glibc)."""
    cap = torch.backends.cpu.get_cpu_capability()
    if cap not in ('AVX512', 'AVX2'):
        pytest.skip(f'host CPU capability {cap}: ATen would not run the Sleef vector loop')
    hal['beam'].set_reference_cpu(cap)
    try:
        gen = torch.Generator().manual_seed(11)
        for n in (1, 7, 33, 528, 1089, 100000 + 19):
            a = torch.randn(n, generator=gen) * 6 - 20
            b = a + torch.randn(n, generator=gen) * torch.tensor([0.01, 1.0, 5.0, 30.0])[torch.randint(0, 4, (n,), generator=gen)]
            a[::97] = float('-inf'); b[::89] = float('-inf')
            want = torch.logaddexp(a, b)
            got = hal['ops'].logaddexp_aten(a.to(DEV), b.to(DEV)).cpu()
            assert torch.equal(got.view(torch.int32), want.view(torch.int32)), n
    finally:
        hal['beam'].set_reference_cpu('AVX512')


def test_topk_replica_matches_torch_cpu(hal):
    """halo_topk_f32 must return torch.topk's CPU order even among equal keys (both ATen code paths)."""
    gen = torch.Generator().manual_seed(0)
    for n, k in [(33, 16), (33, 33), (528, 16), (99, 3), (771, 3), (1089, 33), (2000, 5), (100, 1), (5, 2)]:
        pool = torch.tensor([0., 1., 2., 3., 0.5, float('-inf'), 7.25])
        v = pool[torch.randint(0, len(pool), (40, n), generator=gen)]
        mix = torch.rand(40, n, generator=gen)
        v = torch.where(mix < 0.15, torch.rand(40, n, generator=gen), v)
        want = v.topk(k, dim=1, largest=True, sorted=True)
        vals, idx = hal['ops'].topk(v.to(DEV), k)
        assert torch.equal(idx.cpu(), want.indices), (n, k)
        assert torch.equal(vals.cpu(), want.values), (n, k)


def test_beam_probs_and_errors(hal):
    g = load_golden('g3_beam')
    seqs, scores = hal['beam'].ctc_beam_search_decode_probs(torch.from_numpy(g['probs.probs']).to(DEV), int(g['probs.beam']))
    assert seqs == _unpad(g['probs.seqs'], g['probs.lens'])
    np.testing.assert_allclose(scores.cpu().numpy(), g['probs.scores'], rtol=1e-5)
    with pytest.raises(RuntimeError):
        hal['beam'].ctc_beam_search_decode_logits(torch.zeros(3, 4, device=DEV).log_softmax(-1), beam_size=6)


def test_beam_batch_vs_oracle(hal):
    """Batched entry against the oracle run on THIS host: exact once the kernel is told which vector ISA the host's ATen uses."""
    from oracle import lattice
    cap = torch.backends.cpu.get_cpu_capability()
    hal['beam'].set_reference_cpu(cap)
    try:
        gen = torch.Generator().manual_seed(8)
        em = (torch.randn(6, 21, 32, generator=gen) * 6).log_softmax(-1)
        flat = (torch.randn(3, 21, 32, generator=gen) * 0.5).log_softmax(-1)          # near-flat: score ties decide the survivors
        em = torch.cat([em, flat])
        out, scores = hal['beam'].decode_batch(em.to(DEV), 16, True)
        for n in range(em.shape[0]):
            want_seqs, want_scores = lattice.ctc_beam_search_decode_logits(em[n], 16)
            if cap in ('AVX512', 'AVX2'):
                assert out[n] == want_seqs, n
                np.testing.assert_array_equal(scores[n].cpu().numpy(), want_scores.numpy())
            else:                                          # unknown vector ISA: scores to rounding, best hypothesis' tail
                np.testing.assert_allclose(scores[n].cpu().numpy(), want_scores.numpy(), rtol=1e-6, atol=1e-6)
    finally:
        hal['beam'].set_reference_cpu('AVX512')


# ---------------------------------------------------------------------------- dropout stream parity
def test_philox_mask_equals_oracle(hal):
    from oracle import philox
    n = 100003
    x = torch.ones(n, device=DEV)
    d = hal['ops'].Dropout(0.2, seed=0x123456789abcdef, offset=7)
    y = hal['ops'].dropout_fwd(x, d, 5)
    want = philox.dropout_mask(n, 0.2, 0x123456789abcdef, 5, 7)
    np.testing.assert_array_equal(y.cpu().numpy(), want)
    ctr = torch.tensor([3], dtype=torch.int32, device=DEV)
    d2 = hal['ops'].Dropout(0.2, seed=0x123456789abcdef, offset=4, counter=ctr)
    np.testing.assert_array_equal(hal['ops'].dropout_fwd(x, d2, 5).cpu().numpy(), want)   # 4 + 3 == 7


# -------------------------------------------------------------------------------- the model path
def _load_modules(hal, g, F_, C, H, L, V):
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L)
    rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict({k[len('encoder.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('encoder.')})
    rec.load_state_dict({k[len('recognizer.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('recognizer.')})
    return enc.to(DEV), rec.to(DEV)


@pytest.fixture
def math_mode(request, hal):
    prev = hal['lib'].get_math_mode()
    hal['lib'].set_math_mode(request.param)
    yield request.param
    hal['lib'].set_math_mode(prev)


BOTH_MODES = pytest.mark.parametrize('math_mode', ['f32', 'bf16x3'], indirect=True)


@pytest.fixture
def fusion(request, hal):
    """Layer-diagonal fused LSTM schedule on/off (only takes effect in bf16x3 mode with H % 64 == 0, L >= 2)."""
    hal['lib'].set_lstm_fusion(request.param)
    yield request.param
    hal['lib'].set_lstm_fusion(False)


FUSION = pytest.mark.parametrize('fusion', [False, True], indirect=True)


@BOTH_MODES
@pytest.mark.parametrize('name', ['g1_tiny_l2', 'g1_tiny_l3'])
def test_tiny_model_matches_reference(hal, name, math_mode):
    g = load_golden(name)
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = _load_modules(hal, g, c['F_'], c['C'], c['H'], c['L'], c['V'])
    enc.eval(); rec.eval()
    x, il, tg, tl = (torch.from_numpy(g[k]).to(DEV) for k in ('x', 'il', 'tg', 'tl'))
    feats, flen, _ = enc(x, il)
    assert flen.dtype == torch.int32 and np.array_equal(flen.cpu().numpy(), g['flen'])
    np.testing.assert_allclose(feats.detach().cpu().numpy(), g['feats'], atol=1e-5)
    loss, _ = rec(feats, tg, flen, tl)
    np.testing.assert_allclose(loss.item(), float(g['loss']), rtol=1e-5)
    loss.backward()
    for k, p in list(enc.named_parameters()) + list(rec.named_parameters()):
        key = 'grad.' + ('recognizer.' if k.startswith('classifier') else 'encoder.') + k
        np.testing.assert_allclose(p.grad.cpu().numpy(), g[key], rtol=1e-3, atol=2e-6, err_msg=key)
    with torch.no_grad():
        lp = rec.log_probs(feats)
        np.testing.assert_allclose(lp.cpu().numpy(), g['lp'], atol=1e-5)
        hyps, hlen, ali, scores, none = rec.decode(feats, flen, tl)
    assert none is None and np.array_equal(ali.cpu().numpy(), g['ali']) and np.array_equal(hlen.numpy(), g['hlen'])
    assert [h.tolist() for h in hyps.unbind()] == _unpad(g['hyps'], g['hlen'])


@FUSION
@BOTH_MODES
def test_lc2x1024_matches_reference(hal, math_mode, fusion):
    """BASELINE config 1 shapes: 2-layer H=1024, 80x80 mel, B=4, V=32 against the reference's numbers.
    Both arithmetic modes must meet the fp32-exact tolerances (bf16x3 keeps ~16 bits per operand)."""
    from oracle import cpu_ref
    g = load_golden('g1_lc2x1024')
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
    assert np.array_equal(flen.cpu().numpy(), g['flen'])
    loss.backward()
    np.testing.assert_allclose(feats.grad[:, :, ::61].cpu().numpy(), g['dfeats_slice'], atol=1e-6)
    for k, p in list(enc.named_parameters()) + list(rec.named_parameters()):
        key = ('recognizer.' if k.startswith('classifier') else 'encoder.') + k
        np.testing.assert_allclose(p.grad.double().norm().item(), float(g['gradnorm.' + key]), rtol=1e-4, err_msg=key)
        np.testing.assert_allclose(p.grad.reshape(-1)[::9973].cpu().numpy(), g['gradslice.' + key], rtol=1e-3, atol=1e-6,
                                   err_msg=key)
    with torch.no_grad():
        lp = rec.log_probs(feats)
    np.testing.assert_allclose(lp.cpu().numpy(), g['lp'], atol=1e-4)
    ali, scores, hyp, hlen = hal['ops'].ctc_greedy(lp.contiguous())
    assert np.array_equal(ali.cpu().numpy(), g['ali']) and np.array_equal(hlen.cpu().numpy(), g['hlen'])


def test_lc2x1024_bf16_mode_within_stated_tolerance(hal):
    """HALO_MATH_BF16 (operands of every dense product rounded to bf16, fp32 accumulate and state -- what the reference's
    fp16/bf16 autocast runs compute, ha/loop.py:125): SURVEY.md section 8d's bf16-MFMA tolerance, loss rel <= 2e-2; also
    features <= 3e-2 abs, every gradient within 5 % of its norm and at cosine >= 0.995 on the sampled slice; the greedy
    alignment of this random-init model (near-flat posteriors) agrees on >= 95 % of the frames."""
    from oracle import cpu_ref
    prev = hal['lib'].get_math_mode()
    hal['lib'].set_math_mode('bf16')
    try:
        g = load_golden('g1_lc2x1024')
        c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
        enc_p, rec_p = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], c['seed'])
        x, _, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], c['seed'])
        enc = hal['rnn'].Encoder(c['F_'], c['C'], c['H'], num_layers=c['L'])
        rec = hal['recognizer'].TemporalClassifier(c['H'], c['V'])
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        enc.to(DEV).eval(); rec.to(DEV).eval()
        feats, flen, _ = enc(x.to(DEV), torch.from_numpy(g['il']).to(DEV))
        loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
        assert abs(loss.item() - float(g['loss'])) <= 2e-2 * abs(float(g['loss']))
        assert np.abs(feats[:, :, ::61].detach().cpu().numpy() - g['feats_slice']).max() <= 3e-2
        loss.backward()
        for k, p in list(enc.named_parameters()) + list(rec.named_parameters()):
            key = ('recognizer.' if k.startswith('classifier') else 'encoder.') + k
            want = float(g['gradnorm.' + key])
            assert abs(p.grad.double().norm().item() - want) <= 0.05 * want, key
            a, b = p.grad.reshape(-1)[::9973].cpu().numpy().astype(np.float64), g['gradslice.' + key].astype(np.float64)
            if len(a) >= 16:
                assert a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-30) >= 0.995, key
        with torch.no_grad():
            lp = rec.log_probs(feats)
        ali, _, _, hlen = hal['ops'].ctc_greedy(lp.contiguous())
        assert (ali.cpu().numpy() == g['ali']).mean() >= 0.95
    finally:
        hal['lib'].set_math_mode(prev)


@BOTH_MODES
def test_training_mode_matches_oracle_with_same_masks(hal, math_mode):
    """Dropout on: the HIP path and the CPU restatement consume the same Philox masks."""
    from oracle import cpu_ref
    F_, C, H, L, V, B, T, S = 12, 16, 32, 3, 9, 5, 41, 4
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 21)
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 22)
    Tp = int(cpu_ref.subsampled_lengths(il)[0])
    seed, offset = 99, 3
    masks = cpu_ref.philox_masks(B, Tp, C, H, L, 0.2, 0.2, seed, offset)
    pe = {k: v.clone().requires_grad_(True) for k, v in enc_p.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in rec_p.items()}
    loss_ref, feats_ref, _ = cpu_ref.lstm_ctc_loss(pe, pr, x, il, tg, tl, masks=masks)
    loss_ref.backward()
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).train(); rec.to(DEV).train()
    for m in (enc, rec):
        m.dropout_stream.seed, m.dropout_stream.offset = seed, offset
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))
    loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
    np.testing.assert_allclose(feats.detach().cpu().numpy(), feats_ref.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    loss.backward()
    for k, p in enc.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), pe[k].grad.numpy(), rtol=1e-3, atol=2e-6, err_msg=k)
    for k, p in rec.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), pr[k].grad.numpy(), rtol=1e-3, atol=2e-6, err_msg=k)


@FUSION
@BOTH_MODES
@pytest.mark.parametrize('E', [32, 64])       # 64 with fusion: the layer-diagonal path with carried state
def test_decoder_lm_matches_torch_lstm(hal, math_mode, fusion, E):
    """ha.rnn.Decoder surface: time-major LSTM with carried state + tied output layer."""
    V, L, T, N = 50, 2, 9, 3
    torch.manual_seed(4)
    dec = hal['rnn'].Decoder(V, E, E, L)
    ref = torch.nn.LSTM(E, E, L)
    ref.load_state_dict({k: v for k, v in dec.rnn.state_dict().items()})
    emb_w, out_b = dec.embedding.weight.detach().clone().requires_grad_(True), dec.out_layer.bias.detach().clone()
    tokens = torch.randint(0, V, (T, N))
    h0, c0 = torch.randn(L, N, E) * 0.1, torch.randn(L, N, E) * 0.1
    out_ref, (hn_ref, cn_ref) = ref(torch.nn.functional.embedding(tokens, emb_w), (h0, c0))
    logits_ref = torch.nn.functional.linear(out_ref, emb_w, out_b).view(-1, V)
    dec.to(DEV)
    logits, (hn, cn) = dec(tokens.to(DEV), (h0.to(DEV), c0.to(DEV)))
    np.testing.assert_allclose(logits.detach().cpu().numpy(), logits_ref.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(hn.detach().cpu().numpy(), hn_ref.detach().numpy(), atol=1e-5)
    np.testing.assert_allclose(cn.detach().cpu().numpy(), cn_ref.detach().numpy(), atol=1e-5)
    logits.square().mean().backward()
    logits_ref.square().mean().backward()
    for k, p in dec.rnn.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), getattr(ref, k).grad.numpy(), rtol=1e-3, atol=1e-6, err_msg=k)
    # tied embedding / output weight (ha/rnn.py:42): the HIP gather's scatter-add backward plus the output layer's dW
    np.testing.assert_allclose(dec.embedding.weight.grad.cpu().numpy(), emb_w.grad.numpy(), rtol=1e-3, atol=2e-6)
    lbf, _ = dec.forward_batch_first(tokens.t().to(DEV), (h0.to(DEV), c0.to(DEV)))
    np.testing.assert_allclose(lbf.detach().cpu().numpy(), logits_ref.detach().view(T, N, V).transpose(0, 1).numpy(), atol=1e-5)


@BOTH_MODES
def test_decoder_lm_matches_reference_fixture(hal, math_mode):
    """ha.rnn.Decoder as the reference itself ran it (tests/golden/g10_rnn_decoder.npz: ha/rnn.py:30-77 under the TBPTT pattern of
    rnnlm.py:191-211): two chunks with the carried, detached state, tied output layer, cross-entropy; logits, states, losses and
    every gradient of the second chunk."""
    g = load_golden('g10_rnn_decoder')
    V, E, L, T, N = (int(v) for v in g['cfg'])
    dec = hal['rnn'].Decoder(V, E, E, L)
    dec.load_state_dict({k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}, strict=True)
    dec = dec.to(DEV).eval()
    tokens = torch.from_numpy(g['tokens']).to(DEV)
    state = dec.init_hidden(N)
    assert state[0].shape == (L, N, E) and state[0].device.type == 'cuda'
    tol = 1e-5 if math_mode == 'f32' else 3e-5
    for chunk in range(2):
        dec.zero_grad()
        logits, state = dec(tokens[chunk, :-1], dec.truncate_hidden(state))
        loss = torch.nn.functional.cross_entropy(logits, tokens[chunk, 1:].reshape(-1))
        loss.backward()
        np.testing.assert_allclose(logits.detach().cpu().numpy(), g[f'chunk{chunk}.logits'], atol=tol)
        np.testing.assert_allclose(state[0].detach().cpu().numpy(), g[f'chunk{chunk}.h'], atol=tol)
        np.testing.assert_allclose(state[1].detach().cpu().numpy(), g[f'chunk{chunk}.c'], atol=tol)
        np.testing.assert_allclose(float(loss.detach()), float(g[f'chunk{chunk}.loss']), rtol=1e-5)
    for k, p in dec.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g['grad.' + k], rtol=1e-3, atol=2e-6, err_msg=k)
    with torch.no_grad():
        lbf, _ = dec.forward_batch_first(tokens[0, :-1].t(), dec.init_hidden(N))
    np.testing.assert_allclose(lbf.cpu().numpy(), g['batch_first.logits'], atol=tol)


def test_product_fails_loudly_on_cpu_tensors(hal):
    enc = hal['rnn'].Encoder(12, 16, 32, num_layers=1)
    with pytest.raises(hal['lib'].HaloError):
        enc(torch.randn(2, 20, 12), torch.tensor([20, 20]))


# ------------------------------------------------------------------------- the optimizer step
@BOTH_MODES
@pytest.mark.parametrize('use_graph', [False, True])
def test_train_steps_match_reference(hal, use_graph, math_mode):
    """Three full steps (fwd, CTC, bwd, encoder-only clip 0.1, AdamW with ha/optim.py's decay groups)
    against the reference's own run recorded in g1_train3.npz."""
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    g = load_golden('g1_train3')
    c = {k[4:]: v for k, v in g.items() if k.startswith('cfg_')}
    F_, C, H, L, V = (int(c[k]) for k in ('F_', 'C', 'H', 'L', 'V'))
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, int(c['seed']))
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    tr = LstmCtcTrainer(enc, rec, lr=float(c['lr']), use_graph=use_graph)
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(int(c['B']), int(c['T']), F_, V, int(c['S']), 100 + step)
        loss = tr.step(x.to(DEV), il.to(DEV), tg.to(DEV), tl.to(DEV))
        np.testing.assert_allclose(loss.item(), g['losses'][step], rtol=2e-5)
        np.testing.assert_allclose(tr.grad_norm.item(), g['gnorms'][step], rtol=1e-4)
    sd = {**{'encoder.' + k: v for k, v in enc.state_dict().items()}, **{'recognizer.' + k: v for k, v in rec.state_dict().items()}}
    # Adam normalises by sqrt(v): an element whose gradient is at noise level turns a 1e-7 gradient
    # difference into a visible update difference (lr = 3e-3 here); split-bf16 operands add ~2^-16.
    atol = 5e-6 if math_mode == 'f32' else 2e-5
    for k, v in sd.items():
        np.testing.assert_allclose(v.cpu().numpy(), g['final.' + k], atol=atol, err_msg=k)


def test_trainer_skips_update_on_nonfinite_gradients(hal):
    from oracle import cpu_ref
    from haloop_amd.train import LstmCtcTrainer
    enc_p, rec_p = cpu_ref.make_params(12, 16, 32, 2, 9, 3)
    enc = hal['rnn'].Encoder(12, 16, 32, num_layers=2); rec = hal['recognizer'].TemporalClassifier(32, 9)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    tr = LstmCtcTrainer(enc, rec, use_graph=False)
    x, il, tg, tl = cpu_ref.synthetic_batch(3, 41, 12, 9, 4, 5)
    tg[0] = 1                                       # 4 equal labels need 7 frames, only 3 given -> infeasible
    before = tr.flat.params.clone()
    loss = tr.step(x.to(DEV), torch.tensor([9, 41, 41], device=DEV), tg.to(DEV), torch.tensor([4, 4, 4], device=DEV))
    assert not np.isfinite(loss.item())             # F.ctc_loss gives inf; ha/loop.py:172 skips the batch
    assert torch.equal(tr.flat.params, before)


def test_recognizer_graph_matches_module_path(hal):
    """haloop_amd.infer.LstmCtcRecognizer (one HIP graph) == Encoder + TemporalClassifier.decode."""
    from haloop_amd.infer import LstmCtcRecognizer
    g = load_golden('g1_tiny_l2')
    c = {k[4:]: int(v) for k, v in g.items() if k.startswith('cfg_')}
    enc, rec = _load_modules(hal, g, c['F_'], c['C'], c['H'], c['L'], c['V'])
    reco = LstmCtcRecognizer(enc, rec)
    x = torch.from_numpy(g['x']).to(DEV)
    for _ in range(2):                                   # second call replays the captured graph
        ali, scores, hyp, hlen = reco.recognize(x)
    assert np.array_equal(ali.cpu().numpy(), g['ali']) and np.array_equal(hlen.cpu().numpy(), g['hlen'])
    assert [hyp[i, :n].tolist() for i, n in enumerate(hlen.tolist())] == _unpad(g['hyps'], g['hlen'])


# ------------------------------------------------------------------------------ GPT scoring path
def _gpt_from_golden(hal, name):
    from haloop_amd import attention
    from oracle import gpt_ref
    g = load_golden(name)
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    if name == 'g5_gpt2_small':
        params = gpt_ref.make_gpt_params(vocab, block, n_layer, n_head, n_embd, bool(bias), seed)
    else:
        params = {k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}
    model = attention.GPT(attention.GPTConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head,
                                              n_embd=n_embd, bias=bool(bias), stable_embedding=bool(g.get('stable', 0)),
                                              causal=bool(g.get('causal', 1))))
    model.load_state_dict(params, strict=True)
    assert model.transformer.wte.weight is model.lm_head.weight
    return g, model.to(DEV).eval()


@BOTH_MODES
@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias', 'g5_gpt_tiny_stable', 'g5_gpt_tiny_bidir'])
def test_gpt_tiny_forward_all_matches_reference(hal, name, math_mode):
    g, model = _gpt_from_golden(hal, name)
    inputs, targets = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['targets']).to(DEV)
    with torch.no_grad():
        per_tok = model.forward_all(inputs, targets, reduction='none')
        mean = model.forward_all(inputs, targets, reduction='mean')
    np.testing.assert_allclose(per_tok.cpu().numpy(), g['per_token'], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(mean.item(), float(g['mean']), rtol=1e-5)
    assert np.array_equal(per_tok.cpu().numpy() == 0, g['targets'].reshape(-1) == 0)       # ignore_index=0


@BOTH_MODES
def test_gpt2_small_nats_per_token_matches_reference(hal, math_mode):
    """BASELINE config 3 shape: GPT-2 small, one 1024-token sequence; nats/token vs the CPU reference.
    Tolerance: 1e-4 abs per token, 1e-5 rel on the mean (fp32 path; bf16x3 operands add ~1e-5 per GEMM)."""
    g, model = _gpt_from_golden(hal, 'g5_gpt2_small')
    inputs, targets = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['targets']).to(DEV)
    with torch.inference_mode():
        per_tok = model.forward_all(inputs, targets, reduction='none')
    np.testing.assert_allclose(per_tok.cpu().numpy(), g['per_token'], rtol=0, atol=2e-4)
    valid = g['targets'].reshape(-1) != 0
    np.testing.assert_allclose(per_tok.cpu().numpy()[valid].mean(), float(g['mean']), rtol=2e-5)


def test_gpt_refuses_what_is_not_built(hal):
    from haloop_amd import attention
    cfg = attention.GPTConfig(block_size=16, vocab_size=50, n_layer=1, n_head=1, n_embd=64)
    model = attention.GPT(cfg).to(DEV)
    ids = torch.randint(1, 50, (1, 8), device=DEV)
    with pytest.raises(NotImplementedError):
        model(ids)                                           # generation path under autograd
    with pytest.raises(NotImplementedError):
        attention.GPT(attention.GPTConfig(rotary_emb_dim=32, n_layer=1))   # flash_attn rotary blocks
    with pytest.raises(NotImplementedError):
        with torch.no_grad():
            attention.GPT(attention.GPTConfig(block_size=16, vocab_size=50, n_layer=1, n_head=1, n_embd=64, dropout=0.1)).to(DEV).train() \
                .forward_all(ids, ids)                   # training-mode dropout only exists on the autograd path


# ------------------------------------------------------------------- other shapes of the same path
@FUSION
@BOTH_MODES
@pytest.mark.parametrize('F_,C,H,L,V,B,T,S', [
    (13, 128, 96, 3, 40, 5, 83, 6),        # stock 3-layer encoder on 13 MFCCs (ha/rnn.py:6,11), odd sizes, H % 32 == 0
    (80, 128, 1536, 2, 32, 3, 80, 10),     # the H=1536 variant (ha/init.py:171)
    (20, 24, 48, 1, 11, 17, 37, 5),        # single layer, H % 32 != 0 (f32-packed fallback), B not a multiple of 16
    (20, 32, 64, 3, 11, 20, 45, 5),        # smallest shape of the layer-diagonal fused path (H % 64 == 0, L = 3)
    (40, 64, 128, 2, 32, 64, 80, 10),      # fused path, B = 64
    (40, 64, 256, 2, 32, 8, 400, 12),      # 400-frame utterances: T' = 101 feature frames (beyond the fused head's 32-frame tile)
])
def test_other_shapes_match_oracle(hal, math_mode, fusion, F_, C, H, L, V, B, T, S):
    """Sizes outside the goldens are checked against the CPU oracle (itself pinned to the reference)."""
    from oracle import cpu_ref
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 31)
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 32)
    il = torch.clamp(il - torch.arange(B) % 7, min=T // 2)
    pe = {k: v.clone().requires_grad_(True) for k, v in enc_p.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in rec_p.items()}
    loss_ref, feats_ref, flen_ref = cpu_ref.lstm_ctc_loss(pe, pr, x, il, tg, tl)
    loss_ref.backward()
    enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    enc.to(DEV).eval(); rec.to(DEV).eval()
    feats, flen, _ = enc(x.to(DEV), il.to(DEV))
    loss, _ = rec(feats, tg.to(DEV), flen, tl.to(DEV))
    assert np.array_equal(flen.cpu().numpy(), flen_ref.numpy())
    np.testing.assert_allclose(feats.detach().cpu().numpy(), feats_ref.detach().numpy(), atol=1e-4)
    np.testing.assert_allclose(loss.item(), loss_ref.item(), rtol=1e-5)
    loss.backward()
    for k, p in enc.named_parameters():
        ref = pe[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)
    for k, p in rec.named_parameters():
        ref = pr[k].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-6 + 1e-4 * np.abs(ref).max(), err_msg=k)


def test_hap_scoring_contract(hal):
    """ha/score.py:57-83 semantics on token id lists, against the oracle's forward on the same padded batch."""
    from haloop_amd import attention, score
    from oracle import gpt_ref
    g, model = _gpt_from_golden(hal, 'g5_gpt_tiny_nobias')
    vocab, block, n_layer, n_head = (int(v) for v in g['cfg'][:4])
    gen = torch.Generator().manual_seed(4)
    sents = [torch.randint(1, vocab, (n,), generator=gen).tolist() for n in (5, 17, 40, 1)]     # 40 > block_size 32
    got = score.score_token_batches(model, sents, eos=vocab - 1)
    params = {k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}
    comp = torch.nn.utils.rnn.pad_sequence([torch.LongTensor(s) for s in sents], batch_first=True)[:, :block]
    inp = torch.cat([torch.full((len(sents), 1), vocab - 1), comp[:, :-1]], dim=1)[:, :block]
    with torch.no_grad():
        ref = gpt_ref.gpt_forward_all(params, n_layer, n_head, inp, comp, reduction='none').view(len(sents), -1).sum(-1)
    for (lpt, n, ln), r, s in zip(got, ref.tolist(), sents):
        assert n == min(block, len(s)) and ln == len(s)
        np.testing.assert_allclose(lpt, r / n, rtol=2e-5)
    assert score.format_lines(got)[0].count('\t') == 2


# ------------------------------------------------------------------------------ GPT training direction
@pytest.mark.parametrize('hd,heads,N,Tq,Tk,causal,ragged', [
    (64, 2, 2, 130, 130, True, False), (64, 2, 2, 70, 70, False, True), (32, 3, 2, 9, 100, False, True),
    (16, 2, 3, 5, 5, True, False), (32, 2, 1, 65, 65, True, False), (64, 1, 2, 33, 200, False, False)])
@pytest.mark.parametrize('math_mode', ['f32', 'bf16x3', 'bf16'], indirect=True)
def test_attention_bwd_against_autograd(hal, hd, heads, N, Tq, Tk, causal, ragged, math_mode):
    import math
    ops = hal['ops']
    g = torch.Generator().manual_seed(hd + 3 * Tq + Tk)
    C = heads * hd
    q = torch.randn(N * Tq, C, generator=g, requires_grad=True)
    kv = torch.randn(N * Tk, 2 * C, generator=g, requires_grad=True)
    dy = torch.randn(N * Tq, C, generator=g)
    lens = torch.tensor([Tk - (7 * n) % Tk for n in range(N)], dtype=torch.int32) if ragged else None
    qh = q.view(N, Tq, heads, hd).transpose(1, 2)
    kh = kv[:, :C].reshape(N, Tk, heads, hd).transpose(1, 2)
    vh = kv[:, C:].reshape(N, Tk, heads, hd).transpose(1, 2)
    s = (qh @ kh.transpose(-1, -2)) / math.sqrt(hd)
    if causal:
        s = s.masked_fill(~torch.ones(Tq, Tk, dtype=torch.bool).tril(), float('-inf'))
    if ragged:
        s = s.masked_fill((torch.arange(Tk)[None, :] >= lens[:, None])[:, None, None, :], float('-inf'))
    ref = (s.softmax(-1) @ vh).transpose(1, 2).reshape(N * Tq, C)
    ref.backward(dy)
    qd, kvd, dyd = q.detach().to(DEV), kv.detach().to(DEV), dy.to(DEV)
    ld = lens.to(DEV) if ragged else None
    y, lse, _ = ops.attention_fwd(qd, kvd[:, :C], kvd[:, C:], N, heads, hd, Tq, Tk, causal=causal, key_lengths=ld, want_lse=True)
    dq = torch.full_like(qd, float('nan'))
    dkv = torch.full_like(kvd, float('nan'))
    ops.attention_bwd(qd, kvd[:, :C], kvd[:, C:], y, dyd, lse, dq, dkv[:, :C], dkv[:, C:], N, heads, hd, Tq, Tk, causal=causal,
                      key_lengths=ld)
    tol = {'f32': 2e-5, 'bf16x3': 1e-4, 'bf16': 1e-1}[math_mode]
    np.testing.assert_allclose(dq.cpu().numpy(), q.grad.numpy(), atol=tol, rtol=1e-4)
    np.testing.assert_allclose(dkv.cpu().numpy(), kv.grad.numpy(), atol=tol, rtol=1e-4)


def test_layernorm_gelu_cross_entropy_backward_against_autograd(hal):
    import torch.nn.functional as Fn
    ops = hal['ops']
    g = torch.Generator().manual_seed(11)
    rows, C = 301, 200
    x = torch.randn(rows, C, generator=g, requires_grad=True)
    w = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy, dres = torch.randn(rows, C, generator=g), torch.randn(rows, C, generator=g)
    Fn.layer_norm(x, (C,), w, b, 1e-5).backward(dy)
    dx, dw, db = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), dres.to(DEV), has_bias=True)
    np.testing.assert_allclose(dx.cpu().numpy(), (x.grad + dres).numpy(), atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), atol=5e-5, rtol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), atol=5e-5, rtol=1e-5)
    from oracle import gpt_ref
    for exact in (False, True):
        a = (3 * torch.randn(1000, generator=g)).requires_grad_(True)
        (Fn.gelu(a) if exact else gpt_ref.new_gelu(a)).backward(torch.ones(1000))
        np.testing.assert_allclose(ops.gelu_fwd(a.detach().to(DEV), exact).cpu().numpy(),
                                   (Fn.gelu(a) if exact else gpt_ref.new_gelu(a)).detach().numpy(), atol=1e-6)
        np.testing.assert_allclose(ops.gelu_bwd(torch.ones(1000, device=DEV), a.detach().to(DEV), exact).cpu().numpy(), a.grad.numpy(),
                                   atol=2e-6)
    logits = torch.randn(50, 333, generator=g, requires_grad=True)
    tg = torch.randint(0, 333, (50,), generator=g)
    tg[::7] = 0
    gr = torch.randn(50, generator=g)
    (Fn.cross_entropy(logits, tg, ignore_index=0, reduction='none') * gr).sum().backward()
    ld = logits.detach().to(DEV).clone()
    loss, lse = ops.cross_entropy_fwd_lse(ld, tg.to(DEV))
    ops.cross_entropy_bwd_(ld, tg.to(DEV), lse, gr.to(DEV))
    np.testing.assert_allclose(ld.cpu().numpy(), logits.grad.numpy(), atol=1e-6)


@BOTH_MODES
@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias', 'g5_gpt_tiny_stable', 'g5_gpt_tiny_bidir'])
def test_gpt_tiny_gradients_match_reference(hal, name, math_mode):
    """loss.backward() through the HIP path vs the reference's own gradients (fixtures): every parameter."""
    g, model = _gpt_from_golden(hal, name)
    model.train()
    inputs, targets = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['targets']).to(DEV)
    loss = model.forward_all(inputs, targets, reduction='mean')
    assert loss.requires_grad
    np.testing.assert_allclose(loss.item(), float(g['mean']), rtol=1e-5)
    loss.backward()
    checked = 0
    tol = dict(rtol=2e-4, atol=2e-7) if math_mode == 'f32' else dict(rtol=1e-3, atol=2e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), g['grad.' + k], err_msg=k, **tol)
        checked += 1
    assert checked == sum(1 for k in g if k.startswith('grad.'))
    # gradient accumulation (ha/attention_loop.py:196-208): a second backward adds up
    model.forward_all(inputs, targets, reduction='mean').backward()
    k, p = next(iter(model.named_parameters()))
    np.testing.assert_allclose(p.grad.cpu().numpy(), 2 * g['grad.' + k], rtol=tol['rtol'], atol=2 * tol['atol'])


@BOTH_MODES
def test_gpt2_small_gradients_match_reference(hal, math_mode):
    """GPT-2 small, one 1024-token sequence: norm and a strided 1000-element sample of every parameter's gradient
    against the reference's backward (fixture).  Tolerance: 2e-3 of the gradient's own scale (bf16x3: 5e-3)."""
    g, model = _gpt_from_golden(hal, 'g5_gpt2_small')
    model.train()
    inputs, targets = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['targets']).to(DEV)
    model.forward_all(inputs, targets, reduction='mean').backward()
    rel = 2e-3 if math_mode == 'f32' else 5e-3
    for k, p in model.named_parameters():
        want_norm = float(g['gradnorm.' + k])
        assert abs(float(p.grad.norm()) - want_norm) <= rel * want_norm, k
        sample = p.grad.flatten()[::max(1, p.numel() // 1000)][:1000].cpu().numpy()
        want = g['gradsample.' + k]
        assert np.abs(sample - want).max() <= rel * max(np.abs(want).max(), want_norm / p.numel() ** 0.5), k


@BOTH_MODES
@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias'])
def test_gpt_kv_cache_generation_matches_reference(hal, name, math_mode):
    """GPT.forward(input_ids, past) (ha/attention.py:253-279): prefill + cached continuations vs the reference's logits and
    its `present` cache; forward_all over a cache prefix; generate() yields greedy (top_k=1) tokens equal to the argmax chain."""
    from haloop_amd import attention
    g, model = _gpt_from_golden(hal, name)
    inputs, split = torch.from_numpy(g['inputs']).to(DEV), int(g['gen.split'])
    with torch.no_grad():
        l0, past = model(inputs[:, :split])
        l1, past1 = model(inputs[:, split:split + 1], past=past)
        l2, past2 = model(inputs[:, split + 1:split + 4], past=past1)
        for got, key in ((l0, 'gen.logits0'), (l1, 'gen.logits1'), (l2, 'gen.logits2'), (past2, 'gen.present2')):
            assert got.shape == g[key].shape
            # bf16x3 GEMMs: ~2^-16 relative per product on values of order 1
            np.testing.assert_allclose(got.cpu().numpy(), g[key], rtol=2e-5, atol=1e-5 if math_mode == 'f32' else 1e-4, err_msg=key)
        # per-token loss of a continuation == the same positions of the full pass
        targets = torch.from_numpy(g['targets']).to(DEV)
        T = inputs.shape[1]
        cont = model.forward_all(inputs[:, split:], targets[:, split:], past=past, reduction='none')
        np.testing.assert_allclose(cont.view(-1, T - split).cpu().numpy(), g['per_token'].reshape(-1, T)[:, split:], rtol=2e-5,
                                   atol=2e-5 if math_mode == 'f32' else 1e-4)
        x, present = model.forward_context(inputs)
        assert x.shape == (*inputs.shape, model.config.n_embd) and present.shape[-2] == T
        # greedy generation through the cache equals recomputing from scratch each step
        seq = inputs[:1, :5]
        out = [int(t) for t in attention.generate(model, seq, 4, top_k=1, stop_token=-1)]
        for tok in out:
            logits, _ = model(seq)
            assert int(logits[0, -1].argmax()) == tok
            seq = torch.cat([seq, torch.tensor([[tok]], device=DEV)], dim=1)


def test_gpt_bf16_mode_nats_per_token(hal):
    """HALO_MATH_BF16 (operands rounded to bf16, the arithmetic BASELINE config 3 names): nats/token of GPT-2 small at
    T=1024 within 2e-2 abs of the fp32 CPU reference, gradients within 5 % of their norms."""
    prev = hal['lib'].get_math_mode()
    hal['lib'].set_math_mode('bf16')
    try:
        g, model = _gpt_from_golden(hal, 'g5_gpt2_small')
        inputs, targets = torch.from_numpy(g['inputs']).to(DEV), torch.from_numpy(g['targets']).to(DEV)
        with torch.no_grad():
            per_tok = model.forward_all(inputs, targets, reduction='none').cpu().numpy()
        valid = g['targets'].reshape(-1) != 0
        assert abs(per_tok[valid].mean() - float(g['mean'])) <= 2e-2
        assert np.abs(per_tok - g['per_token']).max() <= 0.25
        model.train()
        model.forward_all(inputs, targets, reduction='mean').backward()
        for k, p in model.named_parameters():
            want = float(g['gradnorm.' + k])
            assert abs(float(p.grad.norm()) - want) <= 0.05 * want, k
    finally:
        hal['lib'].set_math_mode(prev)


@BOTH_MODES
@pytest.mark.parametrize('name', ['g5_gpt_tiny_nobias', 'g5_gpt_tiny_bias'])
def test_gpt_training_mode_dropout_matches_oracle_with_same_masks(hal, name, math_mode):
    """config.dropout = 0.1, model.train(): the Philox masks of every site are restated on the CPU and fed to the oracle."""
    from haloop_amd import attention
    from oracle import gpt_ref, philox
    g = load_golden(name)
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    params = {k[len('param.'):]: torch.from_numpy(v) for k, v in g.items() if k.startswith('param.')}
    P, SEED = 0.1, 0xFEEDFACE12345
    model = attention.GPT(attention.GPTConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head, n_embd=n_embd,
                                              bias=bool(bias), dropout=P))
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).train()
    model.dropout_stream.seed = SEED
    inputs, targets = torch.from_numpy(g['inputs']), torch.from_numpy(g['targets'])
    loss = model.forward_all(inputs.to(DEV), targets.to(DEV))
    loss.backward()
    C = n_embd
    rows = lambda sid: torch.from_numpy(philox.dropout_mask(B * T * C, P, SEED, sid, 0)).view(B, T, C)
    masks = {'emb': rows(64), 'att': [], 'res': [], 'mlp': []}
    for i in range(n_layer):
        masks['att'].append(torch.from_numpy(philox.attention_dropout_mask(B, n_head, T, T, P, SEED, 65 + 3 * i, 0).copy()))
        masks['res'].append(rows(66 + 3 * i))
        masks['mlp'].append(rows(67 + 3 * i))
    pr = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    pr['lm_head.weight'] = pr['transformer.wte.weight']
    ref = gpt_ref.gpt_forward_all(pr, n_layer, n_head, inputs, targets, masks=masks)
    ref.backward()
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
    tol = dict(rtol=5e-4, atol=5e-7) if math_mode == 'f32' else dict(rtol=2e-3, atol=4e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), pr[k].grad.numpy(), err_msg=k, **tol)


@BOTH_MODES
def test_gpt_three_adamw_steps_match_cpu(hal, math_mode):
    """Three optimizer steps (forward_all + backward + halo_adamw) vs the CPU oracle with torch.optim.AdamW: the weight
    operand images cached per parameter version must be rebuilt after every update (a stale image would freeze the
    forward), and the trajectory must follow the reference's."""
    from oracle import gpt_ref
    g, model = _gpt_from_golden(hal, 'g5_gpt_tiny_nobias')
    model.train()
    vocab, block, n_layer, n_head, n_embd, bias, B, T, seed = (int(v) for v in g['cfg'])
    inputs, targets = torch.from_numpy(g['inputs']), torch.from_numpy(g['targets'])
    ref = {k[len('param.'):]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items() if k.startswith('param.')}
    ref['lm_head.weight'] = ref['transformer.wte.weight']
    uniq = [p for k, p in ref.items() if k != 'lm_head.weight']
    opt = torch.optim.AdamW([{'params': [p for p in uniq if p.dim() >= 2], 'weight_decay': 0.1},
                             {'params': [p for p in uniq if p.dim() < 2], 'weight_decay': 0.0}], lr=1e-2, betas=(0.9, 0.95), eps=1e-8)
    params = list(model.parameters())
    state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in params]
    losses, ref_losses = [], []
    for step in range(1, 4):
        for p in params: p.grad = None
        loss = model.forward_all(inputs.to(DEV), targets.to(DEV))
        loss.backward()
        losses.append(loss.item())
        for p, (m, v) in zip(params, state):
            hal['ops'].adamw(p.detach().view(-1), p.grad.view(-1), m.view(-1), v.view(-1), 1e-2, 0.9, 0.95, 1e-8,
                             0.1 if p.dim() >= 2 else 0.0, step)
        opt.zero_grad()
        rl = gpt_ref.gpt_forward_all(ref, n_layer, n_head, inputs, targets)
        rl.backward()
        ref_losses.append(rl.item())
        opt.step()
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-4)
    assert losses[2] < losses[0] - 0.05                    # the forward sees the updated weights
    # Adam normalises each element by its own gradient scale: an element whose gradient is ~0 can take a +-lr step of either
    # sign under last-bit differences, so allow 0.1 % outliers (bounded by 3 steps of lr) on top of the elementwise tolerance
    for k, p in model.named_parameters():
        d = np.abs(p.detach().cpu().numpy() - ref[k].detach().numpy())
        assert (d > (2e-4 if math_mode == 'f32' else 5e-4)).mean() <= 1e-3 and d.max() <= 3.5e-2, k


# ------------------------------------------------------------------------------ token-tape batching (data formats)
@pytest.mark.parametrize('dtype', [torch.uint8, torch.int16, torch.int32, torch.int64])
@pytest.mark.parametrize('n,batch,bptt', [(48, 2, 8), (50, 3, 5), (1000, 7, 33), (5, 8, 4), (4096, 64, 16)])
def test_symbol_tape_batches_bit_exact(hal, dtype, n, batch, bptt):
    from haloop_amd import symbol_tape
    from oracle import tape_ref
    g = torch.Generator().manual_seed(n + batch)
    data = torch.randint(1, 120, (n,), generator=g).to(dtype)
    ref = tape_ref.SymbolTapeNoPad(data.numpy(), batch, bptt)
    tape = symbol_tape.SymbolTapeNoPad(data.to(DEV), batch, bptt)
    assert (len(tape), tape.tape_len, tape.tape_parts, tape.trailing_tokens) == (len(ref), ref.tape_len, ref.tape_parts, ref.trailing_tokens)
    for i in range(len(tape)):
        got = tape[i]
        assert got.dtype == dtype and np.array_equal(got.cpu().numpy(), ref[i]), i
    with pytest.raises(hal['lib'].HaloError):
        symbol_tape.SymbolTapeNoPad(data, batch, bptt)[0]                    # host tape: no CPU path


def test_lm_get_batch_u16_bit_exact(hal, tmp_path):
    from haloop_amd import symbol_tape
    from oracle import tape_ref
    rng = np.random.default_rng(5)
    tokens = rng.integers(0, 50257, size=20000).astype(np.uint16)
    tokens[rng.integers(0, 20000, size=3000)] = 0                            # zeros matter to the "cond" objective
    path = tmp_path / 'tokens.u16'
    tokens.tofile(path)                                                      # the flat u16 file format of ha/spm_encode.py:45-47
    data = symbol_tape.load_u16(path, DEV)
    assert data.numel() == 20000
    offsets = torch.tensor([0, 17, 19000, 19999 - 64, 4242])
    for objective in ('lm', 'cond'):
        x, y = symbol_tape.get_batch(data, offsets, 64, objective)
        xr, yr = tape_ref.get_batch(tokens, offsets.tolist(), 64, objective)
        assert x.dtype == torch.int64 and np.array_equal(x.cpu().numpy(), xr) and np.array_equal(y.cpu().numpy(), yr), objective


@pytest.mark.parametrize('use_graph', [False, True])
def test_gradient_accumulation_equals_one_large_batch(hal, use_graph):
    """--accumulate (ha/loop.py:176-181): two half-batch micro-steps of loss / 2 == one step on the whole batch (dropout off,
    equal halves so the mean-reduced CTC loss is the mean of the halves' means)."""
    from haloop_amd.train import LstmCtcTrainer
    from oracle import cpu_ref
    F_, C, H, L, V, B, T, S = 12, 16, 32, 2, 9, 4, 41, 4
    x, il, tg, tl = (t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 7))

    def build():
        enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 100)
        enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        return enc.to(DEV).eval(), rec.to(DEV).eval()

    enc, rec = build()
    whole = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=False)
    for _ in range(2):
        whole.step(x, il, tg, tl)
    enc2, rec2 = build()
    acc = LstmCtcTrainer(enc2, rec2, lr=3e-3, use_graph=use_graph, accumulate=2)
    for _ in range(2):
        acc.step(x[:2], il[:2], tg[:2], tl[:2])
        assert acc.step_count == _                                  # no update after the first micro-step
        acc.step(x[2:], il[2:], tg[2:], tl[2:])
    assert acc.step_count == 2
    np.testing.assert_allclose(acc.grad_norm.item(), whole.grad_norm.item(), rtol=1e-4)
    np.testing.assert_allclose(acc.flat.params.cpu().numpy(), whole.flat.params.cpu().numpy(), atol=5e-6)


@pytest.mark.parametrize('use_graph', [False, True])
def test_accumulation_recovers_after_a_nonfinite_micro_batch(hal, use_graph):
    """An infeasible utterance (il < tl: F.ctc_loss = inf, NaN gradients) in one micro-batch must not poison the running sum: the
    reference skips that batch and carries on (ha/loop.py:167-174).  Here its contribution is dropped on the device, the cycle's
    update uses the clean micro-batch alone, and the NEXT cycle is an ordinary one.  A cycle whose every micro-batch is bad
    leaves parameters and the Adam step count untouched."""
    from haloop_amd.train import LstmCtcTrainer
    from oracle import cpu_ref
    F_, C, H, L, V, B, T, S = 12, 16, 32, 2, 9, 4, 41, 4
    x, il, tg, tl = (t.to(DEV) for t in cpu_ref.synthetic_batch(B, T, F_, V, S, 7))
    bad_il = il.clone(); bad_il[0] = 9                      # T' = 3 frames for up to 4 labels
    bad_tg = tg.clone(); bad_tg[0] = 1                      # 4 equal labels need 7 frames
    bad_tl = tl.clone(); bad_tl[0] = 4

    def build():
        enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 100)
        enc = hal['rnn'].Encoder(F_, C, H, num_layers=L); rec = hal['recognizer'].TemporalClassifier(H, V)
        enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
        return enc.to(DEV).eval(), rec.to(DEV).eval()

    enc, rec = build()
    tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=use_graph, accumulate=2)
    p0 = tr.flat.params.clone()
    # cycle 1: bad micro-batch first (the case that used to leave NaN in the sum for good), then a clean one
    l1 = tr.step(x[:2], bad_il[:2], bad_tg[:2], bad_tl[:2])
    assert not np.isfinite(l1.item())
    tr.step(x[2:], il[2:], tg[2:], tl[2:])
    assert np.isfinite(tr.grad_norm.item()) and int(tr.adam_step.item()) == 1
    p1 = tr.flat.params.clone()
    assert torch.isfinite(p1).all() and not torch.equal(p1, p0)
    # the same update from a trainer that only ever saw the clean micro-batch at half weight
    enc2, rec2 = build()
    ref = LstmCtcTrainer(enc2, rec2, lr=3e-3, use_graph=False, accumulate=1)
    ref._forward_backward(x[2:], il[2:], tg[2:], tl[2:])
    ref.flat.grads.mul_(0.5)
    ref._optimizer()
    np.testing.assert_allclose(p1.cpu().numpy(), ref.flat.params.cpu().numpy(), atol=2e-6)
    # cycle 2: clean, must update again
    tr.step(x[:2], il[:2], tg[:2], tl[:2]); tr.step(x[2:], il[2:], tg[2:], tl[2:])
    assert int(tr.adam_step.item()) == 2 and not torch.equal(tr.flat.params, p1) and torch.isfinite(tr.flat.params).all()
    # cycle 3: both micro-batches bad -> gradient sum is zero, finite: an (empty) update; parameters move only by decay/momentum.
    # a non-finite NORM (single-batch trainer) leaves everything untouched, including the Adam step count
    enc3, rec3 = build()
    one = LstmCtcTrainer(enc3, rec3, lr=3e-3, use_graph=use_graph)
    q0 = one.flat.params.clone()
    one.step(x[:2], bad_il[:2], bad_tg[:2], bad_tl[:2])
    assert torch.equal(one.flat.params, q0) and int(one.adam_step.item()) == 0
    one.step(x[2:], il[2:], tg[2:], tl[2:])
    assert int(one.adam_step.item()) == 1 and not torch.equal(one.flat.params, q0)


def test_adamw_multi_follows_a_changing_learning_rate(hal):
    """ha/optim.py:68-72 sets the learning rate on every step; weight decay must use the CURRENT lr (torch.optim.AdamW does)."""
    g = torch.Generator().manual_seed(3)
    ps = [torch.randn(300, 7, generator=g), torch.randn(1000, generator=g)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    topt = torch.optim.AdamW([{'params': [ref[0]], 'weight_decay': 0.1}, {'params': [ref[1]], 'weight_decay': 0.0}], lr=1e-2, betas=(0.9, 0.99))
    mine = [torch.nn.Parameter(p.clone().to(DEV)) for p in ps]
    opt = hal['ops'].AdamWMulti(mine, [0.1, 0.0], lr=1e-2, betas=(0.9, 0.99))
    for step, lr in enumerate([1e-2, 3e-1, 5e-2]):
        gs = [torch.randn(p.shape, generator=g) for p in ps]
        for grp in topt.param_groups:
            grp['lr'] = lr
        opt.param_groups[0]['lr'] = lr
        for p, q, gr in zip(ref, mine, gs):
            p.grad = gr.clone(); q.grad = gr.clone().to(DEV)
        topt.step(); opt.step()
    for p, q in zip(ref, mine):
        np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=1e-5, atol=1e-6)


@BOTH_MODES
@pytest.mark.parametrize('vocab,block,n_layer,n_head,n_embd,bias,B,T', [(131, 80, 2, 3, 96, True, 3, 77), (53, 200, 1, 2, 128, False, 1, 193)])
def test_gpt_shape_robustness_against_oracle(hal, math_mode, vocab, block, n_layer, n_head, n_embd, bias, B, T):
    """Sizes the fixtures do not cover (T across several 64-row attention tiles and not a multiple of anything, 3 heads of 32,
    odd vocabulary, B = 1): per-token loss and every gradient against the CPU oracle."""
    from haloop_amd import attention
    from oracle import gpt_ref
    params = gpt_ref.make_gpt_params(vocab, block, n_layer, n_head, n_embd, bias, 11)
    inputs, targets = gpt_ref.synthetic_tokens(B, T, vocab, 12)
    model = attention.GPT(attention.GPTConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head, n_embd=n_embd, bias=bias))
    model.load_state_dict(params, strict=True)
    model = model.to(DEV).train()
    per_tok = model.forward_all(inputs.to(DEV), targets.to(DEV), reduction='none')
    loss = per_tok.sum() / (targets != 0).sum().to(DEV)
    loss.backward()
    ref = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref['lm_head.weight'] = ref['transformer.wte.weight']
    ref_tok = gpt_ref.gpt_forward_all(ref, n_layer, n_head, inputs, targets, reduction='none')
    (ref_tok.sum() / (targets != 0).sum()).backward()
    np.testing.assert_allclose(per_tok.detach().cpu().numpy(), ref_tok.detach().numpy(), rtol=2e-5, atol=5e-5)
    tol = dict(rtol=5e-4, atol=5e-7) if math_mode == 'f32' else dict(rtol=2e-3, atol=4e-6)
    for k, p in model.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), ref[k].grad.numpy(), err_msg=k, **tol)


# ---- one-read operand image pairs (halo_image_pair / halo_cross_entropy_bwd_images) -------------------------------------
@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('R,Cc,pad', [(130, 70, 0), (257, 200, 8), (128, 128, 0), (64, 96, 4)])
def test_image_pair_matches_the_two_single_image_passes(mode, R, Cc, pad):
    """image_pair(x, op) must produce exactly the images split_image makes of op(x) and of op(x)^T: GEMMs fed either way agree
    bit for bit (compared through GEMMs so that the unwritten lo parts of bf16-mode images do not matter)."""
    from haloop_amd import _lib, ops
    _lib.lib()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        g = torch.Generator().manual_seed(R * 1000 + Cc)
        buf = torch.randn(R, Cc + pad, generator=g).cuda()
        x = buf[:, :Cc]                                                     # row stride > cols when pad
        a = torch.randn(R, Cc, generator=g).cuda()
        wb = torch.randn(72, Cc, generator=g).cuda()                        # B operand for the row-major image: [N, K = Cc]
        wc = torch.randn(80, R, generator=g).cuda()                         # B operand for the transposed image: [N, K = R]
        wb_img, wc_img = ops.split_image(wb), ops.split_image(wc)
        cases = [(ops.PAIR_COPY, None, x.contiguous()),
                 (ops.PAIR_GELU, None, ops.gelu_fwd(x.contiguous())),
                 (ops.PAIR_GELU_ERF, None, ops.gelu_fwd(x.contiguous(), exact=True)),
                 (ops.PAIR_GELU_BWD, a, ops.gelu_bwd(x.contiguous(), a)),
                 (ops.PAIR_GELU_ERF_BWD, a, ops.gelu_bwd(x.contiguous(), a, exact=True))]
        for op, x2, value in cases:
            rm, tr = ops.image_pair(x, op, x2)
            ref_rm = ops.gemm_split(ops.split_image(value), wb_img, R, 72, Cc)
            ref_tr = ops.gemm_split(ops.split_image(value, transposed=True), wc_img, Cc, 80, R)
            assert torch.equal(ops.gemm_split(rm, wb_img, R, 72, Cc), ref_rm), op
            assert torch.equal(ops.gemm_split(tr, wc_img, Cc, 80, R), ref_tr), op
            only_rm, none = ops.image_pair(x, op, x2, cols_image=False)
            assert none is None and torch.equal(ops.gemm_split(only_rm, wb_img, R, 72, Cc), ref_rm)
    finally:
        _lib.set_math_mode(prev)


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
def test_cross_entropy_bwd_images_match_the_in_place_gradient(mode):
    from haloop_amd import _lib, ops
    _lib.lib()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        g = torch.Generator().manual_seed(5)
        rows, V, C = 150, 333, 64
        logits = torch.randn(rows, V, generator=g).cuda()
        targets = torch.randint(0, V, (rows,), generator=g).cuda()
        targets[::7] = 0                                                    # ignored rows
        grad = torch.rand(rows, generator=g).cuda()
        loss, lse = ops.cross_entropy_fwd_lse(logits, targets, ignore_index=0)
        w = torch.randn(C, V, generator=g).cuda(); xf = torch.randn(C, rows, generator=g).cuda()
        w_img, xf_img = ops.split_image(w), ops.split_image(xf)
        for gr in (grad, torch.full((1,), 0.25).cuda()):
            rm, tr = ops.cross_entropy_bwd_images(logits, targets, lse, gr, ignore_index=0)
            d = ops.cross_entropy_bwd_(logits.clone(), targets, lse, gr, ignore_index=0)
            assert torch.equal(ops.gemm_split(rm, w_img, rows, C, V), ops.gemm_split(ops.split_image(d), w_img, rows, C, V))
            assert torch.equal(ops.gemm_split(tr, xf_img, V, C, rows), ops.gemm_split(ops.split_image(d, transposed=True), xf_img, V, C, rows))
    finally:
        _lib.set_math_mode(prev)


@pytest.mark.parametrize('rows,C,with_res', [(301, 200, True), (5000, 768, True), (37, 1536, False), (130, 201, True), (9, 64, False),
                                             (2100, 1024, False)])
def test_layernorm_backward_paths(hal, rows, C, with_res):
    """The one-pass fused kernel (C % 4 == 0, <= 1024 and <= 2048 register tiles, rows walked in strides by <= 512 workgroups)
    and the two-pass fallback (C % 4 != 0) against torch autograd."""
    import torch.nn.functional as Fn
    ops = hal['ops']
    g = torch.Generator().manual_seed(rows + C)
    x = (2 * torch.randn(rows, C, generator=g) + 0.5).requires_grad_(True)
    w = (1 + 0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    dy = torch.randn(rows, C, generator=g)
    dres = torch.randn(rows, C, generator=g) if with_res else None
    Fn.layer_norm(x, (C,), w, b, 1e-5).backward(dy)
    dx, dw, db = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), dres.to(DEV) if with_res else None, has_bias=True)
    want_dx = x.grad + dres if with_res else x.grad
    np.testing.assert_allclose(dx.cpu().numpy(), want_dx.numpy(), atol=1e-5, rtol=1e-5)
    scale = float(rows) ** 0.5                                   # the column sums grow like sqrt(rows)
    np.testing.assert_allclose(dw.cpu().numpy(), w.grad.numpy(), atol=2e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(db.cpu().numpy(), b.grad.numpy(), atol=2e-5 * scale, rtol=1e-5)
    dx2, dw2, db2 = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), w.detach().to(DEV), dres.to(DEV) if with_res else None, has_bias=True)
    assert torch.equal(dx, dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)      # fixed summation order


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('M,V,K', [(200, 333, 64), (130, 1000, 96), (70, 64, 128), (257, 50304, 64)])
def test_fused_lm_head_cross_entropy_matches_the_two_step_path(mode, M, V, K):
    """halo_gemm_split_ce (statistics in the GEMM epilogue, strips merged afterwards) against the split GEMM followed by the
    cross-entropy kernel: same logits bit for bit when kept, loss / lse to fp32 rounding; ignored rows give 0."""
    from haloop_amd import _lib, ops
    _lib.lib()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        g = torch.Generator().manual_seed(M + V)
        x = torch.randn(M, K, generator=g).cuda(); w = (0.3 * torch.randn(V, K, generator=g)).cuda()
        bias = (0.1 * torch.randn(V, generator=g)).cuda()
        tg = torch.randint(0, V, (M,), generator=g).cuda()
        tg[::5] = 0
        tg[1] = V - 1                                                        # the last column of the last (ragged) strip
        xi, wi = ops.split_image(x), ops.split_image(w)
        for b in (None, bias):
            ref_logits = ops.gemm_split(xi, wi, M, V, K, bias1=b)
            ref_loss, ref_lse = ops.cross_entropy_fwd_lse(ref_logits, tg, ignore_index=0)
            loss, lse, logits = ops.gemm_split_ce(xi, wi, M, V, K, tg, ignore_index=0, bias=b, want_logits=True, want_lse=True)
            assert torch.equal(logits, ref_logits)
            np.testing.assert_allclose(loss.cpu().numpy(), ref_loss.cpu().numpy(), atol=3e-6, rtol=1e-6)
            np.testing.assert_allclose(lse.cpu().numpy(), ref_lse.cpu().numpy(), atol=3e-6, rtol=1e-6)
            assert float(loss[::5].abs().max()) == 0.0
            loss2, none_lse, none_logits = ops.gemm_split_ce(xi, wi, M, V, K, tg, ignore_index=0, bias=b)
            assert none_lse is None and none_logits is None and torch.equal(loss2, loss)
    finally:
        _lib.set_math_mode(prev)


def test_adamw_multi_is_bit_identical_to_per_tensor_launches():
    """One launch over a parameter list (chunk table, ragged sizes, a tensor smaller than a chunk, one not a multiple of 4) against
    halo_adamw per tensor: identical bits after three steps, with a gradient scale."""
    from haloop_amd import _lib, ops
    _lib.lib()
    g = torch.Generator().manual_seed(3)
    shapes = [(768, 3072), (50304, 16), (768,), (3,), (16384 * 2 + 5,), (130, 7)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    wds = [0.1 if len(s) >= 2 else 0.0 for s in shapes]
    ref_p = [p.detach().clone() for p in params]
    ref_m = [torch.zeros_like(p) for p in ref_p]; ref_v = [torch.zeros_like(p) for p in ref_p]
    opt = ops.AdamWMulti(params, wds, lr=3e-3, betas=(0.9, 0.95), eps=1e-8)
    scale = torch.tensor([0.5], device='cuda')
    for step in range(1, 4):
        grads = [torch.randn(*s, generator=g).cuda() for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr
        opt.step(grad_scale=scale)
        for rp, rm, rv, gr, wd in zip(ref_p, ref_m, ref_v, grads, wds):
            ops.adamw(rp.view(-1), gr.view(-1), rm.view(-1), rv.view(-1), 3e-3, 0.9, 0.95, 1e-8, wd, step, grad_scale=scale)
    for p, rp, m, rm, v, rv in zip(params, ref_p, opt.m, ref_m, opt.v, ref_v):
        assert torch.equal(p.detach(), rp) and torch.equal(m, rm) and torch.equal(v, rv)


@pytest.mark.parametrize('mode', ['bf16x3', 'bf16'])
@pytest.mark.parametrize('M,N,K', [(4100, 3200, 96), (4096, 3072, 32), (3000, 4500, 200)])
def test_split_gemm_many_tiles_ring_variants(mode, M, N, K):
    """Products with >= 768 output tiles take the single-slot three-pass ring (three workgroups per CU) and the lean epilogue
    instantiations: plain, bias + residual-add, and GELU (full epilogue) variants against an fp64 product, ragged edges included."""
    from haloop_amd import _lib, ops
    _lib.lib()
    prev = _lib.get_math_mode()
    _lib.set_math_mode(mode)
    try:
        g = torch.Generator().manual_seed(M + N + K)
        a = torch.randn(M, K, generator=g).cuda(); b = torch.randn(N, K, generator=g).cuda()
        bias = torch.randn(N, generator=g).cuda()
        res = torch.randn(M, N, generator=g).cuda()
        ai, bi = ops.split_image(a), ops.split_image(b)
        ref = a.double() @ b.double().t()
        tol = (3e-5 if mode == 'bf16x3' else 2e-2) * float(ref.abs().max())
        out = ops.gemm_split(ai, bi, M, N, K)
        assert float((out.double() - ref).abs().max()) <= tol
        out = ops.gemm_split(ai, bi, M, N, K, out=res.clone(), bias1=bias, accumulate=True)
        assert float((out.double() - (ref + bias.double() + res.double())).abs().max()) <= tol
        out = ops.gemm_split(ai, bi, M, N, K, bias1=bias, gelu=True)
        want = torch.nn.functional.gelu((ref + bias.double()).float(), approximate='tanh').double()
        assert float((out.double() - want).abs().max()) <= tol
    finally:
        _lib.set_math_mode(prev)


def test_adamw_multi_more_tensors_than_one_launch_carries():
    """300 small parameters: AdamWMulti splits them into launches of at most 256 tensors; still bit-identical to halo_adamw."""
    from haloop_amd import _lib, ops
    _lib.lib()
    g = torch.Generator().manual_seed(8)
    shapes = [(5 + i % 7, 3 + i % 5) if i % 3 else (17 + i,) for i in range(300)]
    params = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    wds = [0.05 if len(s) >= 2 else 0.0 for s in shapes]
    ref_p = [p.detach().clone() for p in params]
    ref_m = [torch.zeros_like(p) for p in ref_p]; ref_v = [torch.zeros_like(p) for p in ref_p]
    opt = ops.AdamWMulti(params, wds, lr=1e-2, betas=(0.9, 0.99), eps=1e-8)
    assert len(opt.groups) == 2
    for step in range(1, 3):
        grads = [torch.randn(*s, generator=g).cuda() for s in shapes]
        for p, gr in zip(params, grads):
            p.grad = gr
        opt.step()
        for rp, rm, rv, gr, wd in zip(ref_p, ref_m, ref_v, grads, wds):
            ops.adamw(rp.view(-1), gr.view(-1), rm.view(-1), rv.view(-1), 1e-2, 0.9, 0.99, 1e-8, wd, step)
    for p, rp in zip(params, ref_p):
        assert torch.equal(p.detach(), rp)


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
@pytest.mark.parametrize('M,N,K,gelu,p', [(300, 200, 96, False, 0.0), (1000, 768, 768, False, 0.0), (256, 128, 2048, False, 0.0),
                                           (300, 200, 96, True, 0.3), (1000, 768, 768, False, 0.3)])
def test_gemm_split_residual_equals_in_place_accumulate(hal, math_mode, M, N, K, gelu, p):
    """halo_gemm_split_residual (the addend of ha/attention.py:178-179's residual connection read from its own buffer) against
    halo_gemm_split's HALO_GEMM_ACCUM into a copy of the addend: BITWISE, on the lean epilogue (bias + add), the full one (GELU /
    dropout) and the split-K path (2 tiles, K = 2048: the addend is copied to the output ahead of the reduce)."""
    ops = hal['ops']
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    bias, r = torch.randn(N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    ai, bi = ops.split_image(a), ops.split_image(b)
    drop = ops.Dropout(p, 77, 3, None) if p else ops.NO_DROPOUT
    want = ops.gemm_split(ai, bi, M, N, K, out=r.clone(), bias1=bias, gelu=gelu, accumulate=True, drop=drop, stream_id=5)
    keep = r.clone()
    got = ops.gemm_split(ai, bi, M, N, K, bias1=bias, gelu=gelu, residual=r, drop=drop, stream_id=5)
    assert torch.equal(got, want) and torch.equal(r, keep)                      # and the addend is left alone
    ref = (a.double() @ b.double().t() + bias.double()).cpu()
    if not gelu and not p:
        tol = 3e-5 if math_mode == 'bf16x3' else 2e-2
        np.testing.assert_allclose(got.cpu().double().numpy(), (ref + r.cpu().double()).numpy(), rtol=0, atol=tol * float(ref.abs().max()))


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
def test_image_pairs_equal_single_launches(hal, math_mode):
    """halo_image_pairs (seven matrices: two launches) against halo_image_pair one by one: the images BITWISE, ragged shapes and a
    row stride wider than the row included (bf16 mode writes the hi parts only: compared through the products that read them)."""
    ops = hal['ops']
    g = torch.Generator().manual_seed(3)
    shapes = [(768, 768), (2304, 768), (100, 70), (129, 33), (64, 3072), (3072, 64), (40, 200)]
    mats = [torch.randn(r, c + (8 if i == 3 else 0), generator=g).to(DEV)[:, :c] for i, (r, c) in enumerate(shapes)]
    got = ops.image_pairs(mats)
    for m, (rm, tr) in zip(mats, got):
        want_rm, want_tr = ops.image_pair(m)
        if math_mode == 'bf16x3':
            assert torch.equal(rm, want_rm) and torch.equal(tr, want_tr)
        R, Cc = m.shape
        x = torch.randn(96, Cc, generator=g).to(DEV)
        xt = torch.randn(96, R, generator=g).to(DEV)
        assert torch.equal(ops.gemm_split(ops.split_image(x), rm, 96, R, Cc), ops.gemm_split(ops.split_image(x), want_rm, 96, R, Cc))
        assert torch.equal(ops.gemm_split(ops.split_image(xt), tr, 96, Cc, R), ops.gemm_split(ops.split_image(xt), want_tr, 96, Cc, R))


@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'], indirect=True)
@pytest.mark.parametrize('M,N,K,N2', [(4096, 1024, 768, 1024), (4000, 1056, 96, 1024), (300, 224, 96, 128)])
def test_gemm_split_io_rowmajor_bf16_on_either_side(hal, math_mode, M, N, K, N2):
    """halo_gemm_split_io: (1) the result written as row-major bf16 is the split of the fp32 result of the same launch, BITWISE
    (hi = bf16(v), lo = bf16(v - hi)), and that fp32 result is halo_gemm_split's (bitwise where that runs without split-K: >= 256
    tiles; to rounding otherwise, and through the GELU epilogue); (2) a product whose A operand is staged from those row-major
    bf16 matrices (source-side swizzle in the LDS-DMA, no operand image) equals the product from the image of the same values, with
    bias and residual add -- ragged M (rows past M are fetched from row M - 1 and never stored)."""
    ops = hal['ops']
    x3 = math_mode == 'bf16x3'
    g = torch.Generator().manual_seed(M + N)
    a, b = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    ai, bi = ops.split_image(a), ops.split_image(b)
    nosplit = ((M + 127) // 128) * ((N + 127) // 128) >= 256

    def same(x, y):
        if nosplit:
            assert torch.equal(x, y)
        else:
            np.testing.assert_allclose(x.cpu().numpy(), y.cpu().numpy(), rtol=0, atol=2e-5 * float(y.abs().max()))

    for gelu in (False, True):
        want = ops.gemm_split(ai, bi, M, N, K, bias1=bias, gelu=gelu)
        c = torch.empty(M, N, device=DEV)
        hi, lo = ops.gemm_split_io(ai, bi, M, N, K, out=c, out_rowmajor=True, bias1=bias, gelu=gelu)
        if gelu:        # the activation's multiply-adds may be contracted differently in the two instantiations: one ulp
            np.testing.assert_allclose(c.cpu().numpy(), want.cpu().numpy(), rtol=2e-6, atol=2e-6)
        else:
            same(c, want)
        assert torch.equal(hi, c.bfloat16())
        if x3:
            assert torch.equal(lo, (c - hi.float()).bfloat16())
        else:
            assert lo is None
        hi2, lo2 = ops.gemm_split_io(ai, bi, M, N, K, out_rowmajor=True, bias1=bias, gelu=gelu)       # bf16 only: no fp32 written
        assert torch.equal(hi2, hi) and (not x3 or torch.equal(lo2, lo))
    # second Linear: A = the bf16 activations of the first, [M, N] x [N2, N]^T with a residual
    hi, lo = ops.gemm_split_io(ai, bi, M, N, K, out_rowmajor=True, bias1=bias)
    act = hi.float() + (lo.float() if x3 else 0.0)                  # what the image of the same values is built from
    w2 = (torch.randn(N2, N, generator=g) / N ** 0.5).to(DEV)
    w2i = ops.split_image(w2)
    r = torch.randn(M, N2, generator=g).to(DEV)
    b2 = torch.randn(N2, generator=g).to(DEV)
    # bf16x3: re-splitting hi + lo gives another (hi, lo) on exact rounding ties (~0.1 % of the elements), so the image path multiplies
    # slightly different operands there; bf16 mode has no such freedom and must agree bitwise
    nosplit = ((M + 127) // 128) * ((N2 + 127) // 128) >= 256 and not x3
    same(ops.gemm_split_io((hi, lo), w2i, M, N2, N, bias1=b2, residual=r), ops.gemm_split(ops.split_image(act), w2i, M, N2, N, bias1=b2, residual=r))
    same(ops.gemm_split_io((hi, lo), w2i, M, N2, N, bias1=b2), ops.gemm_split(ops.split_image(act), w2i, M, N2, N, bias1=b2))
