"""GPU parity of the fused greedy-decode launches (csrc/decode.hip, ha/transformer.py:160-195):

    halo_decode_linear against a float64 product of the same fp32 inputs (split-bf16: <= 3e-5 of the output scale; the LayerNorm and
        GELU variants against F.layer_norm / F.gelu), ragged row counts and feature counts that are not multiples of the 16-wide tile;
    halo_decode_attention_pair against the two launches it replaces (halo_attention_decode + halo_attention_decode_step): outputs
        <= 2e-6 abs (another summation order of the dot products), the fp16 caches BITWISE;
    halo_decode_token against halo_logprob_max + halo_greedy_update + halo_embed_fwd: tokens, alive flags, lengths and the next
        embedding BITWISE, the log-prob / entropy accumulators to 2e-5 + 2e-6 relative (another fixed summation order over the vocabulary);
    Decoder.decode on the fused launches against the operator-per-launch path on a random 3-layer 8x64 model: token ids and lengths
        EXACT, log-probs / entropies <= 2e-3 (the reference-generated `transformer:32` fixture is checked in test_gpu_asr.py, which
        runs the fused path in bf16x3 mode).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def hal():
    from haloop_amd import _lib, ops, transformer
    _lib.lib()
    _lib.lend_scratch()
    return dict(ops=ops, tr=transformer, lib=_lib)


@pytest.mark.parametrize('rows,K,n_out,ln,accumulate,gelu', [
    (64, 512, 2048, True, False, False), (64, 512, 2048, True, False, True), (64, 1024, 512, False, True, False),
    (64, 2048, 512, False, True, False), (50, 512, 32, True, False, False), (7, 768, 40, True, False, False),
    (16, 1024, 100, True, False, True), (33, 1536, 768, False, False, False), (128, 512, 4096, False, True, True)])
def test_decode_linear(hal, rows, K, n_out, ln, accumulate, gelu):
    ops = hal['ops']
    g = torch.Generator().manual_seed(rows * 7 + K + n_out)
    x = torch.randn(rows, K, generator=g) * 1.5 + 0.3
    w = torch.randn(n_out, K, generator=g) / K ** 0.5
    lnw = torch.rand(K, generator=g) + 0.5 if ln else None
    out0 = torch.randn(rows, n_out + 8, generator=g)                                # a wider row buffer: the row stride is honoured
    xin = F.layer_norm(x.double(), (K,), lnw.double(), None, 1e-5) if ln else x.double()
    ref = xin @ w.double().t()
    if gelu:
        ref = F.gelu(ref)
    if accumulate:
        ref = ref + out0[:, :n_out].double()
    assert ops.decode_linear_supported(K, ln)
    img = ops.decode_image(w.to(DEV))
    out = out0.to(DEV)
    ops.decode_linear(x.to(DEV), img, n_out, out, ln_weight=lnw.to(DEV) if ln else None, accumulate=accumulate, gelu=gelu)
    got = out.cpu()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(got[:, :n_out].double().numpy(), ref.numpy(), rtol=0, atol=3e-5 * scale)
    assert torch.equal(got[:, n_out:], out0[:, n_out:])                             # nothing written past the features


@pytest.mark.parametrize('rows,K,n_out', [(64, 1024, 512), (64, 2048, 512), (21, 1024, 100), (128, 2048, 1024)])
def test_decode_linear_with_the_residual_stream_as_a_pair(hal, rows, K, n_out):
    """halo_decode_linear_pair: an accumulating product as two K-slices -- out = (out + side_in) + the first half, side_out = the second
    half -- against the fp64 product; out + side_out is the accumulated stream; the same bits on every run; a LayerNorm launch given the
    pair reads exactly what it reads from the summed rows."""
    ops, lib = hal['ops'], hal['lib']
    g = torch.Generator().manual_seed(rows + K + n_out)
    x = torch.randn(rows, K, generator=g) * 1.5 + 0.3
    w = torch.randn(n_out, K, generator=g) / K ** 0.5
    main0 = torch.randn(rows, n_out + 8, generator=g)
    side0 = torch.randn(rows, n_out + 8, generator=g)
    img = ops.decode_image(w.to(DEV))
    ref = x.double() @ w.double().t() + main0[:, :n_out].double() + side0[:, :n_out].double()
    runs = []
    for _ in range(2):
        main, side_in, side_out = main0.to(DEV), side0.to(DEV), torch.full((rows, n_out + 8), 7.0, device=DEV)
        ops.decode_linear(x.to(DEV), img, n_out, main, accumulate=True, side_in=side_in, side_out=side_out)
        runs.append((main.cpu(), side_out.cpu()))
    (main, side), (main_b, side_b) = runs
    assert torch.equal(main, main_b) and torch.equal(side, side_b)
    got = main[:, :n_out].double() + side[:, :n_out].double()
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))
    assert torch.equal(main[:, n_out:], main0[:, n_out:]) and bool((side[:, n_out:] == 7.0).all())       # nothing written past the features
    # without a side coming in
    main = main0.to(DEV); side_out = torch.zeros(rows, n_out + 8, device=DEV)
    ops.decode_linear(x.to(DEV), img, n_out, main, accumulate=True, side_out=side_out)
    got = (main[:, :n_out].double() + side_out[:, :n_out].double()).cpu()
    np.testing.assert_allclose(got.numpy(), (ref - side0[:, :n_out].double()).numpy(), rtol=0, atol=3e-5 * float(ref.abs().max()))
    # a LayerNorm launch reading the pair == the same launch on the summed rows
    if ops.decode_linear_supported(n_out, True):
        lnw = (torch.rand(n_out, generator=g) + 0.5).to(DEV)
        w2 = torch.randn(64, n_out, generator=g) / n_out ** 0.5
        img2 = ops.decode_image(w2.to(DEV))
        a = main[:, :n_out].contiguous(); b = side_out[:, :n_out].contiguous()
        o1, o2 = torch.zeros(rows, 64, device=DEV), torch.zeros(rows, 64, device=DEV)
        ops.decode_linear(a, img2, 64, o1, ln_weight=lnw, x_side=b)
        ops.decode_linear(a + b, img2, 64, o2, ln_weight=lnw)
        assert torch.equal(o1, o2)
    # refusals: the sliced form needs ACCUM alone and distinct buffers
    with pytest.raises(lib.HaloError):
        ops.decode_linear(x.to(DEV), img, n_out, main, accumulate=False, side_out=side_out)
    with pytest.raises(lib.HaloError):
        ops.decode_linear(x.to(DEV), img, n_out, main, accumulate=True, side_in=side_out, side_out=side_out)


def test_decode_linear_refusals(hal):
    ops, lib = hal['ops'], hal['lib']
    assert not ops.decode_linear_supported(500, False) and not ops.decode_linear_supported(640, True)
    x = torch.zeros(4, 640, device=DEV)
    img = ops.decode_image(torch.zeros(16, 640, device=DEV))
    with pytest.raises(lib.HaloError):
        ops.decode_linear(x, img, 16, torch.zeros(4, 16, device=DEV))


@pytest.mark.parametrize('N,heads,hd,S,T,t', [(5, 8, 64, 10, 9, 0), (5, 8, 64, 10, 9, 4), (3, 2, 32, 70, 80, 71), (64, 8, 64, 10, 9, 8)])
def test_decode_attention_pair_matches_two_launches(hal, N, heads, hd, S, T, t):
    ops = hal['ops']
    C = heads * hd
    g = torch.Generator().manual_seed(N + S + t)
    a = torch.randn(N, 4 * C, generator=g).to(DEV)
    mem = (torch.randn(2, N, heads, S, hd, generator=g)).half().to(DEV)
    time_a = torch.randn(2, N, heads, T, hd, generator=g).half().to(DEV)
    time_a[:, :, :, t:] = 0
    time_b = time_a.clone()
    mlen = torch.randint(1, S + 1, (N,), generator=g, dtype=torch.int32).to(DEV)
    table = ops.RopeTable(T, hd, DEV)
    m = ops.attention_decode(a, mem[0], mem[1], S, key_lengths=mlen)
    s = ops.attention_decode_step(a[:, C:2 * C], a[:, 2 * C:3 * C], a[:, 3 * C:], time_a[0], time_a[1], t + 1, table=table)
    y = torch.full((N, 2 * C), float('nan'), device=DEV)
    ops.decode_attention_pair(a, mem[0], mem[1], mlen, time_b[0], time_b[1], t + 1, table, y)
    # eight lanes per key instead of one: the dot products sum in another order
    np.testing.assert_allclose(y[:, :C].cpu().numpy(), m.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(y[:, C:].cpu().numpy(), s.cpu().numpy(), rtol=0, atol=2e-6)
    assert torch.equal(time_a, time_b)                                               # the fp16 caches: bitwise


def test_decode_memory_caches_equals_per_layer_stores(hal):
    ops = hal['ops']
    L, N, S, heads, hd = 3, 5, 7, 4, 32
    C = heads * hd
    kv = torch.randn(N * S, L * 2 * C + 8, generator=torch.Generator().manual_seed(2)).to(DEV)
    ref = torch.zeros(L, 2, N, heads, S, hd, dtype=torch.float16, device=DEV)
    for l in range(L):
        ops.kv_cache_store(kv[:, l * 2 * C:], C, ref[l, 0], ref[l, 1], N, S, heads, hd, 0)
    got = torch.full_like(ref, float('nan'))
    ops.decode_memory_caches(kv, got)
    assert torch.equal(got, ref)


@pytest.mark.parametrize('N,V,C,t,plen', [(64, 32, 512, 0, 0), (64, 32, 512, 3, 0), (10, 100, 64, 0, 1), (130, 300, 128, 2, 0), (300, 32, 64, 1, 0)])
def test_decode_token_equals_three_launches(hal, N, V, C, t, plen):
    ops = hal['ops']
    g = torch.Generator().manual_seed(N + V + t)
    logits = (torch.randn(N, V, generator=g) * 3).to(DEV)
    wte = torch.randn(V, C, generator=g).to(DEV)
    tokens = torch.randint(0, V, (N, 8), generator=g).to(DEV)
    alive = (torch.rand(N, generator=g) < 0.7).to(torch.uint8).to(DEV)
    state = [torch.randint(0, 5, (N,), generator=g, dtype=torch.int32).to(DEV), torch.randn(N, generator=g).to(DEV),
             torch.randn(N, generator=g).to(DEV)]
    ETX = 3
    t_a, al_a, st_a = tokens.clone(), alive.clone(), [s.clone() for s in state]
    val, idx, ne = ops.logprob_max(logits, want_entropy=True)
    ops.greedy_update(val, idx, ne, t_a, t, plen, ETX, al_a, *st_a)
    y_a = ops.embed_fwd(t_a[:, t + 1:t + 2], wte, None)
    t_b, st_b = tokens.clone(), [s.clone() for s in state]
    al_b = torch.full((2, N), 7, dtype=torch.uint8, device=DEV)                      # step t reads plane t & 1, writes the other
    al_b[t & 1] = alive
    y_b = torch.empty(N, C, device=DEV)
    ops.decode_token(logits, t_b, t, plen, ETX, al_b, *st_b, wte, y_b)
    assert torch.equal(al_b[t & 1], alive) and torch.equal(al_a, al_b[(t + 1) & 1])
    assert torch.equal(t_a, t_b) and torch.equal(y_a.view(N, C), y_b)
    assert torch.equal(st_a[0], st_b[0])                                             # output lengths
    for u, v in zip(st_a[1:], st_b[1:]):    # log-probs, entropies: the sums over the vocabulary run in another (fixed) order
        np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=2e-6, atol=2e-5)


@pytest.mark.parametrize('prompt', [False, True])
def test_fused_decode_matches_operator_path(hal, monkeypatch, prompt):
    from oracle import transformer_ref
    tr, lib = hal['tr'], hal['lib']
    prev = lib.get_math_mode()
    lib.set_math_mode('bf16x3')
    try:
        V, hd, heads, L, N, S = 32, 64, 8, 3, 21, 10
        pd = transformer_ref.make_decoder_params(V, hd, heads, L, 11)
        dec = tr.Decoder(vocab=V, head_dim=hd, heads=heads, p_drop=0.2, layers=L)
        dec.load_state_dict({k[len('decoder.'):]: v for k, v in pd.items() if k.startswith('decoder.')}, strict=True)
        dec = dec.to(DEV).eval()
        g = torch.Generator().manual_seed(5)
        feats = torch.randn(N, S, heads * hd, generator=g).to(DEV)
        flen = torch.randint(3, S + 1, (N,), generator=g).to(DEV)
        tl = torch.randint(4, 9, (N,), generator=g).to(DEV)
        pr = torch.tensor([[7, 9]] * N) if prompt else None
        assert dec._fused_decode_ok(heads * hd)
        res = {}
        for fused in ('1', '0'):
            monkeypatch.setenv('HALO_DECODE_FUSED', fused)
            for graph in ('1', '0'):
                monkeypatch.setenv('HALO_DECODE_GRAPH', graph)
                with torch.no_grad():
                    outs, olen, _, lps, ents = dec.decode(feats, flen, tl, prompt=pr)
                res[fused, graph] = ([o.tolist() for o in outs.unbind()], olen.cpu(), lps.cpu(), ents.cpu())
        base = res['0', '0']
        for key, (toks, olen, lps, ents) in res.items():
            assert toks == base[0] and torch.equal(olen, base[1]), key
            np.testing.assert_allclose(lps.numpy(), base[2].numpy(), rtol=0, atol=2e-3, err_msg=str(key))
            np.testing.assert_allclose(ents.numpy(), base[3].numpy(), rtol=1e-4, atol=2e-3, err_msg=str(key))
        assert torch.equal(res['1', '1'][2], res['1', '0'][2])                       # graph replay = eager launches, bitwise
    finally:
        lib.set_math_mode(prev)
