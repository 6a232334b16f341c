"""The ranged kernels of the sharded data-parallel update (csrc/optim.hip: halo_sumsq_ranges, halo_pack_ranges_bf16, halo_expand_ranges_bf16)
against their definitions in torch, and dp.SpanSharded's device paths against its CPU paths on one rank."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def ops():
    from haloop_amd import _lib, ops
    _lib.lib()
    return ops


def test_sumsq_ranges_is_the_sum_of_squares_of_the_concatenation(ops):
    from haloop_amd import _lib
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1 << 20, generator=g).to(DEV)
    for ranges in ([(0, 1 << 20)], [(4, 1000), (4096, 4096 + 262144)], [(8, 12), (100, 104), (1000, 70000), (500000, 1 << 20)],
                   [(0, 4)] * 1 + [(4 * i, 4 * i + 4) for i in range(10, 17)]):
        parts = torch.full((_lib.HALO_SUMSQ_PARTS,), float('nan'), device=DEV)
        ops.sumsq_ranges(x, ranges, parts)
        want = sum(float((x[a:b].double() ** 2).sum()) for a, b in ranges)
        np.testing.assert_allclose(float(parts.double().sum()), want, rtol=1e-6)
    # one range: bit for bit the partials of the plain launch
    a = ops.sumsq_partials(x[256:256 + 65536])
    b = ops.sumsq_ranges(x, [(256, 256 + 65536)], torch.empty_like(a))
    assert torch.equal(a, b)
    with pytest.raises(_lib.HaloError):
        ops.sumsq_ranges(x, [(2, 10)], parts)              # not on a float4


def test_pack_and_expand_ranges_bf16(ops):
    g = torch.Generator().manual_seed(4)
    world, rank = 4, 2
    lo = [64, 64 + 4 * 96]                                   # two spans of `world` chunks each
    chunks = [96, 40]
    n = lo[1] + world * chunks[1] + 32
    params = torch.randn(n, generator=g).to(DEV)
    own = [(lo[k] + rank * chunks[k], lo[k] + (rank + 1) * chunks[k]) for k in range(2)]
    mine = torch.empty(sum(chunks), dtype=torch.bfloat16, device=DEV)
    ops.pack_ranges_bf16(params, own, mine)
    want = torch.cat([params[a:b] for a, b in own]).to(torch.bfloat16)
    assert torch.equal(mine, want)
    # the gathered buffer: rank-major records; every chunk but this rank's is overwritten with the record's values
    stage = torch.randn(world * sum(chunks), generator=g).to(torch.bfloat16).to(DEV)
    before = params.clone()
    ops.expand_ranges_bf16(stage, lo, chunks, world, rank, params)
    exp = before.clone()
    per = sum(chunks)
    off = 0
    for k in range(2):
        for r in range(world):
            if r != rank:
                exp[lo[k] + r * chunks[k]:lo[k] + (r + 1) * chunks[k]] = stage[r * per + off:r * per + off + chunks[k]].float()
        off += chunks[k]
    assert torch.equal(params, exp)
    assert torch.equal(params[:64], before[:64]) and torch.equal(params[-32:], before[-32:])       # nothing outside the spans moved
    # skip_rank = -1: every chunk
    ops.expand_ranges_bf16(stage, lo, chunks, world, -1, params)
    for k, a in enumerate(own):
        o = sum(chunks[:k])
        assert torch.equal(params[a[0]:a[1]], stage[rank * per + o:rank * per + o + chunks[k]].float())


def test_flat_layout_puts_the_big_matrices_last_in_readiness_order():
    from haloop_amd import rnn, recognizer
    from haloop_amd.train import FlatParams
    enc = rnn.Encoder(20, 16, 32, num_layers=3).to(DEV)
    rec = recognizer.TemporalClassifier(32, 9).to(DEV)
    f = FlatParams(enc, rec)
    names = [n for n, _, _ in f.slots]
    big = [n for n in names if n.startswith('encoder.lstm.weight_') and (n.split('weight_')[1].startswith('hh') or not n.endswith('_l0'))]
    assert names[-len(big):] == ['encoder.lstm.weight_hh_l0', 'encoder.lstm.weight_hh_l1', 'encoder.lstm.weight_ih_l1',
                                 'encoder.lstm.weight_hh_l2', 'encoder.lstm.weight_ih_l2']
    off = {n: o for n, _, o in f.slots}
    assert f.big_late == (off['encoder.lstm.weight_hh_l0'], off['encoder.lstm.weight_hh_l2'])
    assert f.big_early == (off['encoder.lstm.weight_hh_l2'], f.total)
    assert f.small_range == (0, f.big_late[0]) and f.encoder_range == (off['encoder.subsample.bias'], f.total)
    # every parameter is a view of the flat buffer, the AdamW ranges tile it, and (decay, clip) classes are what ha/optim.py:84-106 says
    assert sorted((a, b) for a, b, _, _ in f.ranges) == [(a, b) for a, b, _, _ in f.ranges] and f.ranges[0][0] == 0 and f.ranges[-1][1] == f.total
    for (a, b, decays, clipped) in f.ranges:
        for n, p, o in f.slots:
            if a <= o < b:
                assert clipped == n.startswith('encoder.'), n
                assert decays == (n.startswith('encoder.lstm.') or not n.endswith('bias')), n
