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
CFG_PERSIST = dict(F_=20, C=64, H=256, L=2, V=11, B=6, T=45, S=5)      # H = 256: the persistent recurrence (and its DP toggle) runs


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build(seed, c=None):
    from haloop_amd import rnn, recognizer
    from oracle import cpu_ref
    c = c or CFG
    enc_p, rec_p = cpu_ref.make_params(c['F_'], c['C'], c['H'], c['L'], c['V'], seed)
    enc = rnn.Encoder(c['F_'], c['C'], c['H'], num_layers=c['L']); rec = recognizer.TemporalClassifier(c['H'], c['V'])
    enc.load_state_dict(enc_p); rec.load_state_dict(rec_p)
    return enc.to('cuda:0').eval(), rec.to('cuda:0').eval()


def _worker(rank, world, port, out, use_graph, cfg=None, grad_dtype='f32', dp_algo='allreduce', math_mode=None):
    import datetime
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(110, exit=True)       # a rank stuck in a GPU or gloo call ends itself (with its stacks on stderr)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        if math_mode:
            from haloop_amd import _lib
            _lib.set_math_mode(math_mode)
        _worker_body(rank, world, out, use_graph, cfg or CFG, grad_dtype, dp_algo)
    except Exception:                                   # a dead rank must not leave its peer (or pytest) waiting
        out.put(('error', rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


def _worker_body(rank, world, out, use_graph, c, grad_dtype, dp_algo):
    if True:
        from haloop_amd import dp
        from haloop_amd.train import LstmCtcTrainer
        from oracle import cpu_ref
        enc, rec = _build(100 + rank, c)                   # different init per rank: rank 0's must win
        # (gather_dtype: the trainer's default is 'f32' -- exact parameters on every rank after every step; 'auto' opts into the bf16
        #  all-gather in bf16 arithmetic, which this test exercises together with the exchange of the masters)
        tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=use_graph, grad_dtype=grad_dtype, dp_algo=dp_algo, gather_dtype='auto')
        assert tr.dp_algo == dp_algo
        assert LstmCtcTrainer.__init__.__kwdefaults__ is None and 'f32' in LstmCtcTrainer.__init__.__defaults__
        if dp_algo in ('rs_ag', 'direct'):
            from haloop_amd import _lib
            assert isinstance(tr.sharded, dp.SpanSharded) and tr.sharded.gather_bf16 == (_lib.get_math_mode() == 'bf16')
            assert isinstance(tr.sharded, dp.DirectExchange) == (dp_algo == 'direct')
            if dp_algo == 'direct':         # the small range's all-reduce: one exchange up to three ranks, two from four on
                assert (tr.sharded._small_chunk > 0) == (world >= 4)
        x, il, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7)
        sl = dp.shard_slice(c['B'], rank, world)
        for _ in range(2):
            tr.step(x[sl].cuda(), il[sl].cuda(), tg[sl].cuda(), tl[sl].cuda())
        assert tr.masters_stale == bool(getattr(tr.sharded, 'gather_bf16', False))
        sd = tr.state_dict()                # (after a bf16 all-gather: exchanges the other rank's fp32 master values first)
        assert not tr.masters_stale and set(sd) == {'encoder', 'recognizer'}
        tr.check_status()                   # (a direct exchange that gave up a wait raised the status word)
        torch.cuda.synchronize()
        if rank == 0:
            out.put(('ok', tr.flat.params[:tr.flat.total].cpu().numpy(), float(tr.grad_norm.item())))


@pytest.mark.timeout(600)
@pytest.mark.parametrize('use_graph,cfg_name,grad_dtype,dp_algo,math_mode', [
    (False, 'tiny', 'f32', 'allreduce', None), (True, 'tiny', 'f32', 'allreduce', None),      # True: three captured graphs, all-reduces between them
    (True, 'persist', 'f32', 'allreduce', None), (True, 'persist', 'bf16', 'allreduce', None),
    # the flat sharded update: reduce-scatter, each rank clips (global norm) and updates its half of the flat parameters, all-gather
    (False, 'tiny', 'f32', 'rs_ag_flat', None), (True, 'tiny', 'f32', 'rs_ag_flat', None), (True, 'persist', 'f32', 'rs_ag_flat', None),
    # the span-sharded update (dp.SpanSharded): matrix spans reduce-scattered (over gloo: one reduce per owner), the small parameters
    # all-reduced and updated by both ranks, fp32 all-gather -- and in bf16 arithmetic the bf16 all-gather through the staging buffer
    (False, 'tiny', 'f32', 'rs_ag', None), (True, 'tiny', 'f32', 'rs_ag', None), (True, 'persist', 'f32', 'rs_ag', None),
    (True, 'persist', 'f32', 'rs_ag', 'bf16'), (False, 'persist', 'f32', 'rs_ag', 'bf16'),
    # the same cut with the library's own exchange (dp.DirectExchange): two PROCESSES on one GPU, each writing its pieces into the other's
    # HIP-IPC-mapped arena (epoch words, bounded waits); fp32 gather, and the bf16 gather of bf16 arithmetic
    (False, 'tiny', 'f32', 'direct', None), (True, 'persist', 'f32', 'direct', None), (False, 'persist', 'f32', 'direct', 'bf16')])
def test_two_ranks_equal_single_process_on_concatenated_batch(use_graph, cfg_name, grad_dtype, dp_algo, math_mode):
    _ranks_equal_single_process(2, use_graph, cfg_name, grad_dtype, dp_algo, math_mode)


# four PROCESSES on one GPU: the direct exchange with three peers per rank, and the small range's all-reduce in its two-exchange form
# (pieces to their owners, rank-order sums, summed pieces to everybody: chosen from four ranks on)
@pytest.mark.timeout(600)
@pytest.mark.parametrize('cfg_name,math_mode', [('tiny', None), ('persist', 'bf16')])
def test_four_ranks_equal_single_process_with_the_direct_exchange(cfg_name, math_mode):
    _ranks_equal_single_process(4, False, cfg_name, 'f32', 'direct', math_mode)


def _ranks_equal_single_process(world, use_graph, cfg_name, grad_dtype, dp_algo, math_mode):
    from haloop_amd.train import LstmCtcTrainer
    from oracle import cpu_ref
    cfg = CFG if cfg_name == 'tiny' else CFG_PERSIST
    if cfg['B'] % world:
        cfg = dict(cfg, B=(cfg['B'] + world - 1) // world * world)        # equal shards: the mean of the ranks' means is the batch mean
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, use_graph, cfg, grad_dtype, dp_algo, math_mode)) for r in range(world)]
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
    c = cfg
    from haloop_amd import _lib
    prev_mode = _lib.get_math_mode()
    if math_mode:
        _lib.set_math_mode(math_mode)
    enc, rec = _build(100, c)
    tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=False)
    x, il, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7)
    for _ in range(2):
        tr.step(x.cuda(), il.cuda(), tg.cuda(), tl.cuda())
    _lib.set_math_mode(prev_mode)
    if grad_dtype == 'bf16':          # gradients rounded to 8 significant bits on the wire: the update direction survives, not its bits
        np.testing.assert_allclose(gnorm2, tr.grad_norm.item(), rtol=2e-2)
        diff = np.abs(params2 - tr.flat.params[:tr.flat.total].cpu().numpy())
        # two Adam steps at lr = 3e-3: an element whose tiny gradient changes sign on the wire moves the other way (<= 4 lr apart)
        assert diff.max() <= 4.2 * 3e-3 and (diff > 2e-3).mean() < 1e-3, (diff.max(), (diff > 2e-3).mean())
    else:
        np.testing.assert_allclose(gnorm2, tr.grad_norm.item(), rtol=2e-2 if math_mode == 'bf16' else 1e-4)
        if math_mode == 'bf16':
            # single-pass bf16 products: the batch split changes which fp32 sums the bf16-rounded state sees, and Adam's normalised update
            # carries a flipped last bit of a tiny gradient into the weights (<= 2 lr per step): most elements agree closely, none is far
            diff = np.abs(params2 - tr.flat.params[:tr.flat.total].cpu().numpy())
            assert diff.max() <= 4.2 * 3e-3 and (diff > 1e-4).mean() < 2e-2, (diff.max(), (diff > 1e-4).mean())
        elif cfg_name == 'tiny':
            np.testing.assert_allclose(params2, tr.flat.params[:tr.flat.total].cpu().numpy(), atol=5e-6)
        else:       # H = 256 products run split-bf16 with shape-dependent K slicing: the two batch splits round differently, and Adam's
                    # normalised update carries that into a few near-zero-gradient elements (11 of 865 k above 2e-5 when this was set)
            diff = np.abs(params2 - tr.flat.params[:tr.flat.total].cpu().numpy())
            assert diff.max() <= 1e-4 and (diff > 2e-5).mean() < 1e-4, (diff.max(), (diff > 2e-5).mean())


# ---- the reference's own multi-process path: DistributedDataParallel around GPT (ha/attention_loop.py:152-155,203) ----
def _gpt_worker(rank, world, port, out):
    import datetime
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(110, exit=True)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        from torch.nn.parallel import DistributedDataParallel as DDP
        from haloop_amd import attention
        from oracle import gpt_ref
        torch.cuda.set_device(0)
        torch.manual_seed(100 + rank)                       # different init per rank: DDP broadcasts rank 0's
        model = attention.GPT(attention.GPTConfig(block_size=32, vocab_size=61, n_layer=2, n_head=2, n_embd=64)).to('cuda:0').train()
        class LM(torch.nn.Module):                          # DDP only hooks what runs under its own forward()
            def __init__(self, gpt):
                super().__init__()
                self.gpt = gpt

            def forward(self, x, y):
                return self.gpt.forward_all(x, y)

        ddp = DDP(LM(model), device_ids=[0])
        inputs, targets = gpt_ref.synthetic_tokens(4, 24, 61, 9, pad_tail=False)
        sl = slice(rank * 2, rank * 2 + 2)
        loss = ddp(inputs[sl].cuda(), targets[sl].cuda())
        loss.backward()
        torch.cuda.synchronize()
        if rank == 0:
            out.put(('ok', {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()},
                     {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}))
    except Exception:
        out.put(('error', rank, traceback.format_exc()))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_gpt_under_torch_ddp_averages_gradients():
    """The hand-written GPT backward hands its gradients to autograd like any Function, so torch's DDP (what `hala` wraps the
    model in) all-reduces them: two ranks on half batches == one process on the whole batch."""
    from haloop_amd import attention
    from oracle import gpt_ref
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_gpt_worker, args=(r, 2, port, out)) for r in range(2)]
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
    _, state, grads2 = msg
    model = attention.GPT(attention.GPTConfig(block_size=32, vocab_size=61, n_layer=2, n_head=2, n_embd=64))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    model = model.to('cuda:0').train()
    inputs, targets = gpt_ref.synthetic_tokens(4, 24, 61, 9, pad_tail=False)
    model.forward_all(inputs.cuda(), targets.cuda()).backward()
    for k, p in model.named_parameters():
        # two half-batch sums averaged against one whole-batch sum: fp32 regrouping, ~1e-4 of the gradients' 1e-2 scale on their small elements
        np.testing.assert_allclose(grads2[k], p.grad.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def _rccl_worker(port, out, grad_dtype):
    """ONE rank on the real `nccl` (= RCCL) backend with the trainer's world>1 code path forced on: the broadcast, the AVG probe, the
    three-graph step with the asynchronous all-reduces between the replays, the bf16 wire format.  An all-reduce over one rank is the
    identity, so with ReduceOp.AVG the trajectory must equal the plain single-process trainer's."""
    import datetime
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    try:
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device('cuda', 0))
        from haloop_amd import dp
        from haloop_amd.train import LstmCtcTrainer
        from oracle import cpu_ref
        c = CFG_PERSIST
        x, il, tg, tl = (t.cuda() for t in cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7))
        res = {}
        for forced in (False, True):
            real = dp.world_size
            if forced:
                dp.world_size = lambda group=None: 2              # take every world > 1 branch; the collectives still span one rank
            try:
                enc, rec = _build(100, c)
                tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=True, grad_dtype=grad_dtype if forced else 'f32', dp_algo='allreduce')
                if forced:
                    assert tr.world == 2 and len(tr.avg_early.buckets) >= 1
                    if grad_dtype == 'bf16':                       # SUM over one rank, then the 1/world scale: undo the forced halving
                        tr.avg_early.world = tr.avg_late.world = 1
                        tr.avg_early.start = lambda s=tr.avg_early: [s.reduce_bucket(i, async_op=True) for i in range(len(s.buckets))]
                        tr.avg_late.start = lambda s=tr.avg_late: [s.reduce_bucket(i, async_op=True) for i in range(len(s.buckets))]
                        fin = lambda s: (lambda works: ([w.wait() for w in works], [s._from_wire(i) for i in range(len(s.buckets))]))
                        tr.avg_early.finish, tr.avg_late.finish = fin(tr.avg_early), fin(tr.avg_late)
                losses = []
                for _ in range(3):
                    losses.append(float(tr.step(x, il, tg, tl).item()))
                torch.cuda.synchronize()
                res[forced] = (tr.flat.params.cpu().numpy(), losses, len(tr._graphs), bool(getattr(tr.avg_early, '_avg_op', False)))
            finally:
                dp.world_size = real
        out.put(('ok', res))
    except Exception:
        out.put(('error', traceback.format_exc()))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize('grad_dtype', ['f32', 'bf16'])
def test_rccl_backend_runs_the_data_parallel_step(grad_dtype):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out, grad_dtype))
    p.start()
    try:
        msg = out.get(timeout=300)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg[0] == 'ok', msg[1]
    single, forced = msg[1][False], msg[1][True]
    assert single[2] == 1 and forced[2] == 3                           # one graph / three graphs with the collectives between them
    assert all(np.isfinite(forced[1]))
    if grad_dtype == 'f32' and forced[3]:                              # ReduceOp.AVG over one rank: the identity
        # the lower layers run on the step-launch chain beside the collective (another summation order of the recurrence)
        np.testing.assert_allclose(forced[1], single[1], rtol=1e-5)
        d = np.abs(forced[0] - single[0])
        assert d.max() <= 1e-4 and (d > 2e-5).mean() < 1e-4
    else:                                                              # bf16 wire: gradients rounded to bf16 once
        np.testing.assert_allclose(forced[1][0], single[1][0], rtol=1e-5)
        np.testing.assert_allclose(forced[1], single[1], rtol=2e-2)


def _rccl_sharded_worker(port, out, math_mode):
    """ONE rank on the real `nccl` (= RCCL) backend running the sharded data-parallel step (dp_algo 'rs_ag') with every collective issued
    over that one rank: reduce_scatter_tensor and all_gather_into_tensor in place on the flat buffers, the all-reduce of the norm partials,
    and -- from the second step on -- all three captured in a HIP graph with the two optimizer pieces between them."""
    import datetime
    import faulthandler
    import traceback
    faulthandler.dump_traceback_later(150, exit=True)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    try:
        torch.cuda.set_device(0)
        dist.init_process_group('nccl', rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device('cuda', 0))
        from haloop_amd import _lib, dp
        from haloop_amd.train import LstmCtcTrainer
        from oracle import cpu_ref
        _lib.set_math_mode(math_mode)
        c = CFG_PERSIST
        x, il, tg, tl = (t.cuda() for t in cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7))
        res = {}
        for rehearse in (False, True):
            enc, rec = _build(100, c)
            tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=True, rehearse_dp=rehearse, gather_dtype='auto')
            assert (tr.sharded is not None) == rehearse
            if rehearse:
                assert isinstance(tr.sharded, dp.SpanSharded) and tr.sharded._native and tr.sharded.gather_bf16 == (math_mode == 'bf16')
            losses = [float(tr.step(x, il, tg, tl).item()) for _ in range(4)]
            torch.cuda.synchronize()
            tr.check_status()
            res[rehearse] = (tr.flat.params[:tr.flat.total].cpu().numpy(), losses, getattr(tr, '_tail_graph', None) is not None,
                             bool(getattr(tr, '_early_started', False)))
            if rehearse:
                # the step's pieces bracketed by events (bench.py --gpus N prints them): every piece there, their sum = the step
                comp = tr.profile_dp_components(x, il, tg, tl, steps=2)
                want = {'forward_backward', 'wait_early_reduce_scatter', 'reduce_scatter_early_tail', 'reduce_scatter_late', 'all_reduce_small',
                        'norm_partials_and_their_all_reduce', 'clip_and_adamw_on_owned_ranges', 'all_gather', 'step_total'}
                assert set(comp) == want and all(v >= 0 for v in comp.values()), comp
                assert abs(sum(v for k, v in comp.items() if k != 'step_total') - comp['step_total']) <= 0.02 * comp['step_total'] + 1.0, comp
                tr.check_status()
        # the same with bf16 on the wire of the reduce-scatter (cast, reduce_scatter_tensor on the bf16 buffer, cast back: captured too)
        enc, rec = _build(100, c)
        tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=True, rehearse_dp=True, grad_dtype='bf16')
        assert tr.sharded is not None and tr.sharded._wire is not None
        losses = [float(tr.step(x, il, tg, tl).item()) for _ in range(4)]
        torch.cuda.synchronize()
        tr.check_status()
        res['bf16'] = (tr.flat.params[:tr.flat.total].cpu().numpy(), losses, getattr(tr, '_tail_graph', None) is not None)
        out.put(('ok', res))
    except Exception:
        out.put(('error', traceback.format_exc()))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(400)
@pytest.mark.parametrize('math_mode', ['bf16x3', 'bf16'])
def test_rccl_backend_runs_the_sharded_step(math_mode):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=_rccl_sharded_worker, args=(_free_port(), out, math_mode))
    p.start()
    try:
        msg = out.get(timeout=300)
    finally:
        p.join(60)
        if p.is_alive():
            p.kill()
    assert msg[0] == 'ok', msg[1]
    plain, sharded = msg[1][False], msg[1][True]
    # over one rank every collective is the identity and the span is the whole buffer: the same launches on the same data
    assert sharded[1] == plain[1]
    assert np.array_equal(sharded[0], plain[0])
    assert sharded[2], 'the collectives were not captured (they ran eagerly): see the warning in the log'
    # bf16 arithmetic at this shape runs the two-layer launches: the top layer's reduce-scatter was started from the library's event, on
    # the side stream, in the middle of the backward; elsewhere the tail reduces that span itself
    assert sharded[3] == (math_mode == 'bf16'), sharded[3]
    # bf16 on the wire: the first step's loss is the same (same weights), the trajectory follows within the gradients' bf16 rounding
    wire = msg[1]['bf16']
    assert wire[2] and wire[1][0] == plain[1][0]
    np.testing.assert_allclose(wire[1], plain[1], rtol=2e-2)
    assert not np.array_equal(wire[0], plain[0])


def test_bench_line_of_two_ranks_sharing_the_gpu():
    """bench.py's whole N > 1 path -- it starts its own ranks, gloo rendezvous, the direct peer exchange over HIP IPC, max-over-ranks timing,
    dp_components_us -- rehearsed with both ranks on this one GPU (--share-gpu): the line's contract, not a scaling number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--share-gpu', '--dp-algo', 'direct', '--steps', '4',
                        '--warmup', '2'], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # rank 0 alone prints
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['n_ranks_seen'] == 2 and d['steps'] == 4 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 2 * d['config']['batch_per_gpu'] and d['config']['dp_algo'] == 'direct'
    assert 'REHEARSAL' in d['config']['parallelism']
    assert abs(d['value'] - d['config']['global_batch'] / (d['ms_per_step'] * 1e-3)) <= 1e-3 * d['value']
    comp = d['config']['dp_components_us']
    for k in ('forward_backward', 'reduce_scatter_late', 'all_reduce_small', 'clip_and_adamw_on_owned_ranges', 'all_gather', 'step_total'):
        assert comp[k] > 0
    assert np.isfinite(d['final_loss'])
