"""Two ranks through the real HIP path on one GPU (gloo as the transport, both ranks on cuda:0):
LstmCtcTrainer's broadcast + gradient averaging must reproduce the single-process step on the
concatenated batch (SURVEY.md section 8e parity check, scaled to what one box allows)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(F_=12, C=16, H=32, L=2, V=9, B=4, T=41, S=4)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build(seed):
    from haloop_amd import rnn, recognizer
    from oracle import cpu_ref
    c = CFG
    enc_p, rec_p = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], seed)
    enc = rnn.Encoder(c['F_'], c['C'], c['H'], num_layers=c['L']); rec = recognizer.TemporalClassifier(c['H'], c['V'])
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    return enc.to('cuda:0').eval(), rec.to('cuda:0').eval()


def _worker(rank, world, port, out, use_graph):
    import datetime
    import traceback
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        _worker_body(rank, world, out, use_graph)
    except Exception:                                   # a dead rank must not leave its peer (or pytest) waiting
        out.put(('error', rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, out, use_graph):
    if True:
        from haloop_amd import dp
        from haloop_amd.train import LstmCtcTrainer
        from oracle import cpu_ref
        c = CFG
        enc, rec = _build(100 + rank)                      # different init per rank: rank 0's must win
        tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=use_graph)
        x, il, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7)
        sl = dp.shard_slice(c['B'], rank, world)
        for _ in range(2):
            tr.step(x[sl].cuda(), il[sl].cuda(), tg[sl].cuda(), tl[sl].cuda())
        torch.cuda.synchronize()
        if rank == 0:
            out.put(('ok', tr.flat.params.cpu().numpy(), float(tr.grad_norm.item())))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('use_graph', [False, True])     # True: two captured graphs with the all-reduce between them
def test_two_ranks_equal_single_process_on_concatenated_batch(use_graph):
    from haloop_amd.train import LstmCtcTrainer
    from oracle import cpu_ref
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, use_graph)) for r in range(2)]
    for p in procs:
        p.start()
    import time
    deadline, msg = time.time() + 150, None
    while msg is None and time.time() < deadline:
        if not out.empty():
            msg = out.get()
        elif not any(p.is_alive() for p in procs):
            break
        else:
            time.sleep(0.2)
    for p in procs:
        p.join(20)
        if p.is_alive():
            p.kill()
    assert msg is not None, 'workers produced no result'
    assert msg[0] == 'ok', msg
    _, params2, gnorm2 = msg
    c = CFG
    enc, rec = _build(100)
    tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=False)
    x, il, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7)
    for _ in range(2):
        tr.step(x.cuda(), il.cuda(), tg.cuda(), tl.cuda())
    np.testing.assert_allclose(gnorm2, tr.grad_norm.item(), rtol=1e-4)
    np.testing.assert_allclose(params2, tr.flat.params.cpu().numpy(), atol=5e-6)
