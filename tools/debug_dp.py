#!/usr/bin/env python3
"""Debug helper: the two-rank LstmCtcTrainer step on one GPU (gloo), with progress markers written unbuffered."""
import os, sys, time, socket, datetime, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def log(*a):
    print(f'[{time.time() % 1000:8.2f}]', *a, flush=True)


def worker(rank, world, port, use_graph, persist, grad_dtype):
    try:
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        log(rank, 'init pg')
        dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        log(rank, 'pg ok')
        import test_gpu_dp as T
        from haloop_amd import dp
        from haloop_amd.train import LstmCtcTrainer
        from oracle import cpu_ref
        c = T.CFG_PERSIST if persist else T.CFG
        enc, rec = T._build(100 + rank, c)
        log(rank, 'built')
        tr = LstmCtcTrainer(enc, rec, lr=3e-3, use_graph=use_graph, grad_dtype=grad_dtype)
        log(rank, 'trainer ok')
        x, il, tg, tl = cpu_ref.synthetic_batch(c['B'], c['T'], c['F_'], c['V'], c['S'], 7)
        sl = dp.shard_slice(c['B'], rank, world)
        for i in range(2):
            tr.step(x[sl].cuda(), il[sl].cuda(), tg[sl].cuda(), tl[sl].cuda())
            log(rank, 'step issued', i)
            torch.cuda.synchronize()
            log(rank, 'step done', i)
        dist.destroy_process_group()
        log(rank, 'exit')
    except Exception:
        log(rank, 'ERROR', traceback.format_exc())


if __name__ == '__main__':
    use_graph = 'eager' not in sys.argv
    persist = 'persist' in sys.argv
    grad_dtype = 'bf16' if 'bf16' in sys.argv else 'f32'
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]
    ctx = mp.get_context('spawn')
    procs = [ctx.Process(target=worker, args=(r, 2, port, use_graph, persist, grad_dtype)) for r in range(2)]
    for p in procs: p.start()
    t0 = time.time()
    while time.time() - t0 < 90 and any(p.is_alive() for p in procs):
        time.sleep(0.5)
    for p in procs:
        if p.is_alive():
            log('killing', p.pid); p.kill()
    log('main done')
