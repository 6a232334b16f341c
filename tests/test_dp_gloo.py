"""Data-parallel semantics on CPU: world_size 2, gloo.  The compute here is the CPU oracle (the HIP
path needs a GPU); what is under test is haloop_amd.dp -- broadcast, sharding and gradient
averaging -- and that an N-rank step equals the single-process step on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from haloop_amd import dp
        from oracle import cpu_ref
        torch.set_num_threads(1)
        F_, C, H, L, V, B, T, S = 12, 16, 32, 2, 9, 4, 41, 4
        # every rank draws its own init (like seed 1337+rank, attention_loop.py:75); rank 0's must win
        enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 100 + rank)
        names = [('e', k) for k in enc_p] + [('r', k) for k in rec_p]
        tensors = [enc_p[k] if w == 'e' else rec_p[k] for w, k in names]
        flat = torch.cat([t.reshape(-1) for t in tensors]).clone()
        dp.broadcast_parameters(flat)
        off, pe, pr = 0, {}, {}
        for (w, k), t in zip(names, tensors):
            v = flat[off:off + t.numel()].view(t.shape).clone().requires_grad_(True)
            (pe if w == 'e' else pr)[k] = v
            off += t.numel()
        x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 7)
        sl = dp.shard_slice(B, rank, world)
        loss, _, _ = cpu_ref.lstm_ctc_loss(pe, pr, x[sl], il[sl], tg[sl], tl[sl])
        loss.backward()
        grads = torch.cat([(pe[k] if w == 'e' else pr[k]).grad.reshape(-1) for w, k in names])
        g16 = grads.clone()
        avg = dp.GradientAverager(grads, bucket_bytes=4096, boundaries=[1000, 5000])
        assert len(avg.buckets) > 3
        avg.average()
        avg16 = dp.GradientAverager(g16, bucket_bytes=4096, boundaries=[1000, 5000], wire_dtype='bf16')     # bf16 on the wire
        avg16.average()
        # the sharded update (dp.ShardedUpdate): reduce-scatter the gradient, clip by the GLOBAL norm (partials summed over the ranks),
        # a plain SGD step on this rank's span only, all-gather the parameters
        n = flat.numel()
        npad = dp.ShardedUpdate.padded_numel(n, world)
        P, G = torch.zeros(npad), torch.zeros(npad)
        P[:n] = flat
        G[:n] = torch.cat([(pe[k] if w == 'e' else pr[k]).grad.reshape(-1) for w, k in names])     # this rank's own gradient again
        sh = dp.ShardedUpdate(P, G)
        lo, hi = sh.span
        assert hi - lo == npad // world and sh.rank == rank
        sh.reduce_scatter()
        part = torch.tensor([float((G[lo:hi].double() ** 2).sum())], dtype=torch.float64)
        sh.all_reduce_sum(part)
        coef = min(1.0, 0.05 / (float(part.sqrt()) + 1e-6))
        P[lo:hi] -= 0.1 * coef * G[lo:hi]
        other = (lo - 1) % npad if lo else hi            # an element of another rank's span: untouched until the all-gather
        before = float(P[other])
        sh.all_gather()
        # the same reduce-scatter with bf16 on the wire: this rank's span of the mean, through bf16 sums
        G16 = torch.zeros(npad)
        G16[:n] = torch.cat([(pe[k] if w == 'e' else pr[k]).grad.reshape(-1) for w, k in names])
        sh16 = dp.ShardedUpdate(P.clone(), G16, wire_dtype='bf16')
        sh16.reduce_scatter()
        span16 = torch.zeros(npad)
        span16[lo:hi] = G16[lo:hi]
        dist.all_reduce(span16)                               # (test plumbing: collect every rank's span on rank 0)
        if rank == 0:
            out.put((flat.numpy(), grads.numpy(), float(loss), g16.numpy(), P[:n].numpy(), float(part.sqrt()), before != float(P[other]),
                     span16[:n].numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_on_concatenated_batch():
    from oracle import cpu_ref
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    flat, grads, _, g16, p_sharded, norm_sharded, moved, rs16 = out.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    F_, C, H, L, V, B, T, S = 12, 16, 32, 2, 9, 4, 41, 4
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, 100)              # rank 0's init
    want_flat = torch.cat([t.reshape(-1) for t in list(enc_p.values()) + list(rec_p.values())]).numpy()
    np.testing.assert_array_equal(flat, want_flat)                        # broadcast from rank 0
    pe = {k: v.clone().requires_grad_(True) for k, v in enc_p.items()}
    pr = {k: v.clone().requires_grad_(True) for k, v in rec_p.items()}
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, 7)
    loss, _, _ = cpu_ref.lstm_ctc_loss(pe, pr, x, il, tg, tl)
    loss.backward()
    want = torch.cat([v.grad.reshape(-1) for v in list(pe.values()) + list(pr.values())]).numpy()
    np.testing.assert_allclose(grads, want, rtol=1e-4, atol=1e-7)         # mean of shard means == global mean
    # the bf16 wire format: each rank's contribution rounded to 8 significant bits, summed in bf16, averaged in fp32
    np.testing.assert_allclose(g16, want, rtol=2e-2, atol=1e-2 * np.abs(want).max())
    assert np.abs(g16 - want).max() > 0                                   # it really went through bf16
    # the sharded update == the single-process clipped step on the concatenated batch
    norm = float(np.sqrt((want.astype(np.float64) ** 2).sum()))
    np.testing.assert_allclose(norm_sharded, norm, rtol=1e-4)
    np.testing.assert_allclose(p_sharded, want_flat - 0.1 * min(1.0, 0.05 / (norm + 1e-6)) * want, rtol=1e-4, atol=1e-7)
    assert moved                                                          # the other rank's span arrived through the all-gather
    # the sharded reduce-scatter with bf16 on the wire: the mean to bf16 accuracy, and not the fp32 one
    np.testing.assert_allclose(rs16, want, rtol=2e-2, atol=1e-2 * np.abs(want).max())
    assert np.abs(rs16 - want).max() > 0


def test_shard_slice_and_buckets():
    from haloop_amd import dp
    assert dp.shard_slice(64, 3, 8) == slice(24, 32)
    with pytest.raises(ValueError):
        dp.shard_slice(10, 0, 4)
    g = torch.zeros(1000)
    a = dp.GradientAverager(g, bucket_bytes=400, boundaries=[250])
    assert a.buckets[0] == (0, 100) and (200, 250) in a.buckets and a.buckets[-1][1] == 1000
    assert sum(b - a_ for a_, b in a.buckets) == 1000
    a.average()                                                           # world 1: no-op, no process group needed


def _span_worker(rank, world, port, out, gather_bf16):
    """dp.SpanSharded over gloo: per-span reduce-scatter (one reduce per owner: only the owner receives the sum), the small range
    all-reduced, a clipped SGD step on what the rank owns, all-gather (fp32 in place, or bf16 roundings through the staging buffer)."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from haloop_amd import dp
        torch.set_num_threads(1)
        small, late, early = (0, 104), (104, 104 + 64 * world), (104 + 64 * world, 104 + 64 * world + 96 * world)
        n = early[1]
        g = torch.Generator().manual_seed(5)
        P0 = torch.randn(n, generator=g)                                   # the same parameters on every rank (as after the broadcast)
        G_all = torch.randn(world, n, generator=g)                         # every rank's own gradient
        P, G = P0.clone(), G_all[rank].clone()
        sh = dp.SpanSharded(P, G, early, late, small, gather_bf16=gather_bf16)
        assert sh.world == world and sh.rank == rank and not sh._native
        own = sh.own_ranges()
        assert (small in own) and len(own) == 3 and sum(b - a for a, b in own) == 104 + 64 + 96
        before = G.clone()
        assert sh.reduce_scatter('early') is None
        sh.reduce_scatter('late')
        sh.all_reduce_small()
        # a real reduce-scatter: the chunks this rank does not own still hold its OWN gradient (an all-reduce would have summed them)
        for name in ('early', 'late'):
            lo, hi, c = sh.spans[name]
            for r in range(world):
                if r != rank:
                    assert torch.equal(G[lo + r * c:lo + (r + 1) * c], before[lo + r * c:lo + (r + 1) * c]), (name, r)
        part = torch.tensor([sum(float((G[a:b].double() ** 2).sum()) for a, b in sh.norm_ranges())], dtype=torch.float64)
        sh.all_reduce_sum(part)
        coef = min(1.0, 0.5 / (float(part.sqrt()) + 1e-6))
        for a, b in own:
            P[a:b] -= 0.1 * coef * G[a:b]
        sh.all_gather()
        rounded = P.clone()
        sh.gather_masters()
        if rank == 0:
            out.put((P0.numpy(), G_all.numpy(), rounded.numpy(), P.numpy(), float(part.sqrt()), [sh.own('early'), sh.own('late')], sh.wire_bytes()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('gather_bf16', [False, True])
def test_span_sharded_update_equals_the_single_process_step(gather_bf16):
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_span_worker, args=(r, world, port, out, gather_bf16)) for r in range(world)]
    for p in procs:
        p.start()
    P0, G_all, rounded, masters, norm, owned, wire = out.get()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    mean = G_all.astype(np.float64).mean(0)
    want_norm = float(np.sqrt((mean ** 2).sum()))
    np.testing.assert_allclose(norm, want_norm, rtol=1e-6)                 # the small range counted once, the chunks once each
    want = P0 - 0.1 * min(1.0, 0.5 / (want_norm + 1e-6)) * mean
    np.testing.assert_allclose(masters, want, rtol=1e-5, atol=1e-6)        # after gather_masters: the single-process step, every element
    if gather_bf16:
        mine = np.zeros(len(P0), bool)
        mine[:104] = True
        for a, b in owned:
            mine[a:b] = True
        np.testing.assert_array_equal(rounded[mine], masters[mine])          # own chunks and the small range: fp32 masters
        others = torch.from_numpy(masters[~mine]).to(torch.bfloat16).float().numpy()
        np.testing.assert_array_equal(rounded[~mine], others)                # the other rank's chunks: exactly their bf16 roundings
        assert wire['all_gather'] == 2 * (64 + 96) * world // 2
    else:
        np.testing.assert_array_equal(rounded, masters)
        assert wire['all_gather'] == 4 * (64 + 96) * world // 2
    assert wire['reduce_scatter_early'] == 4 * 96 * world // 2 and wire['reduce_scatter_late'] == 4 * 64 * world // 2


def test_flat_params_layout_on_cpu():
    """FlatParams: every tensor 16-byte aligned and inside exactly one AdamW range; the small (replicated) range is a prefix whose length is
    a multiple of 64 elements (up to 16 ranks cut it into equal 16-byte-aligned pieces: dp.DirectExchange), its padding never belongs to a
    parameter; the late / early spans are the lower layers' and the top LSTM layer's matrices; unexpected LSTM parameter names (a
    bidirectional or projected nn.LSTM) fall into the small range instead of raising."""
    from haloop_amd import rnn, recognizer
    from haloop_amd.train import FlatParams
    enc = rnn.Encoder(12, 16, 32, num_layers=3); rec = recognizer.TemporalClassifier(32, 9)
    f = FlatParams(enc, rec, pad_to=16)
    assert f.small_range[0] == 0 and f.small_range[1] % 64 == 0 and f.small_range[1] == f.big_late[0] and f.big_late[1] == f.big_early[0]
    assert f.big_early[1] == f.total and f.padded % 16 == 0 and f.padded >= f.total
    H = 32
    assert f.big_early[1] - f.big_early[0] == 2 * 4 * H * H                       # the top layer's W_ih and W_hh
    assert f.big_late[1] - f.big_late[0] == 4 * H * H + 2 * 4 * H * H             # layer 0's W_hh, layer 1's W_ih and W_hh
    covered = torch.zeros(f.padded, dtype=torch.int32)
    for name, p, off in f.slots:
        assert off % 4 == 0
        covered[off:off + p.numel()] += 1
        inside = [r for r in f.ranges if r[0] <= off and off + p.numel() <= r[1]]
        assert len(inside) == 1, name
        big = name.startswith('encoder.lstm.weight_') and not name.endswith('ih_l0')
        assert (off >= f.small_range[1]) == big, name
    assert int(covered.max()) == 1
    assert int(covered[f.small_range[1]:f.total].min()) == 1                      # no padding inside the matrix spans
    # the parameters are views of the flat buffer
    name, p, off = f.slots[0]
    assert p.data_ptr() == f.params[off:].data_ptr()

    class Odd(torch.nn.Module):                                                   # names the layout code must tolerate
        def __init__(self):
            super().__init__()
            self.subsample = torch.nn.Conv1d(12, 16, 5)
            self.lstm = torch.nn.LSTM(16, 32, num_layers=2, bidirectional=True, proj_size=8)
    f2 = FlatParams(Odd(), rec, pad_to=4)
    names = {n for n, _, off in f2.slots if off < f2.small_range[1]}
    assert any(n.endswith('_reverse') for n in names) and any('weight_hr' in n for n in names)
