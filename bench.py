#!/usr/bin/env python3
"""Headline benchmark: utterances/sec of the LSTM-CTC training step (BASELINE.json configs[1]).

Workload: 2-layer H=1024 LSTM-CTC ("LC-2x1024"), 64 utterances of 80 frames x
80 mels per GPU, char vocab 32, dropout 0.2 on, full step = forward + CTC loss + backward +
encoder-only clip + AdamW (ha/loop.py:176-196).  Synthetic inputs resident in HBM.  Arithmetic of the
headline number: `bf16` -- the operands of the dense products (input / recurrent / classifier GEMMs)
rounded to bf16, one MFMA per product, fp32 accumulate; cell state, gates, CTC lattice, softmax, clip and
AdamW in fp32 on fp32 master weights: the precision the contract names (the reference's own GPU runs
use fp16 autocast, ha/loop.py:125) at BASELINE.md's bf16-MFMA parity gate (loss rel <= 2e-2; tested at
this very configuration, dropout on, against the CPU oracle with the same masks:
tests/test_gpu_lstm_b64.py).  The same step in the fp32-grade split-bf16 arithmetic (`bf16x3_mode`:
three MFMAs per product, meets the fp32 tolerances) and on the exact-f32 MFMA (`f32_mode`) is
reported beside it, as are larger per-GPU batches (`b_sweep`).

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no torch.distributed environment, this process starts
`python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD (before anything here has
touched the GPU) and forwards its JSON line and exit code; under torchrun it is one of the N ranks.

Prints ONE JSON line on rank 0 (contract in the task statement), including
  roofline     -- the dominant kernel (the recurrent chain of the backward: in bf16 mode ONE persistent launch
                  for both layers, otherwise one per layer, or T' step launches on the fallback path)
                  against the HBM roof; bytes follow SURVEY.md 8d (parameters once per pass + per-step
                  activations), duration measured here with HIP events recorded by the library on the
                  launch stream;
  gpt2_small, asr_transformer32 -- BASELINE configs 3 and 5 (tools/bench_gpt.py, tools/bench_asr.py run as
                  child processes after the timed region, bounded);
  cpu_baseline -- the CPU restatement of the reference path (oracle/, kind "port") timed on this
                  box's host cores on a bounded sample of the same workload (B=64, and B=4 = config 1).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_PER_GPU, T, F, C_SUB, H, L, V, S = 64, 80, 80, 128, 1024, 2, 32, 10
T_SUB = (T + 6 - 5) // 4 + 1   # 21
MATH_DTYPE = {'f32': 'f32', 'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMAs per product, fp32 accumulate)',
              'bf16': 'bf16 (dense operands rounded to bf16, 1 MFMA per product, fp32 accumulate; fp32 state, CTC, optimizer and master weights)'}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='launch the step eagerly')
    ap.add_argument('--graph', action='store_true',
                    help='replay the step from its HIP graph.  Neither flag (one process): both ways are timed over 40 steps after the warm-up and the '
                         'faster one runs the timed region -- at 13 launches per step the host keeps ahead of the GPU and eager launches save the '
                         '~15 us a graph replay costs on ROCm 7.2; a slow or busy host reverses that')
    ap.add_argument('--no-extras', action='store_true', help='skip the inference / bf16_mode / f32_mode legs')
    ap.add_argument('--dp-rehearsal', action='store_true',
                    help='N=1 only, a measurement aid: run the data-parallel code path (three graphs, RCCL all-reduces between them, '
                         'step-launch chain under the first bucket) on a one-rank RCCL group; the line says so in config.parallelism')
    ap.add_argument('--share-gpu', action='store_true',
                    help='N > 1, a rehearsal aid for a one-GPU box: every rank runs on cuda:0 and the process group is gloo (RCCL refuses two '
                         'ranks on one device).  Exercises the whole N > 1 path of this file -- rendezvous, sharded step, collectives or the '
                         'direct peer exchange over HIP IPC, max-over-ranks timing, dp_components_us -- with the ranks sharing one GPU: the '
                         'value is NOT a scaling number and the line says so in config.parallelism')
    ap.add_argument('--math', choices=['f32', 'bf16x3', 'bf16'], default='bf16',
                    help="arithmetic of the dense products: 'bf16' rounds the operands to bf16 (one MFMA per product); 'f32' and "
                         "'bf16x3' meet the fp32 parity tolerances")
    ap.add_argument('--no-configs', action='store_true', help='skip the GPT-2 small / attention-ASR legs (BASELINE configs 3 and 5)')
    ap.add_argument('--dp-algo', choices=['allreduce', 'rs_ag', 'rs_ag_flat', 'direct'], default='rs_ag',
                    help='N > 1: allreduce = every rank averages the whole gradient and updates every parameter (DistributedDataParallel); '
                         'rs_ag = the span-sharded step: the top LSTM layer\'s matrix gradients reduce-scattered from the middle of the backward, the '
                         'lower layers\' behind it, small parameters all-reduced, each rank updates its chunks, bf16 all-gather in bf16 arithmetic; '
                         'rs_ag_flat = one reduce-scatter / fp32 all-gather over the whole flat buffers; direct = the cut of rs_ag with this '
                         'library\'s own exchange over HIP-IPC-mapped peer buffers (each rank writes its pieces straight into the owners\' memory: '
                         'csrc/dp_direct.hip) instead of RCCL collectives')
    ap.add_argument('--gather-dtype', choices=['auto', 'f32', 'bf16'], default='auto',
                    help="N > 1, rs_ag: wire format of the parameter all-gather.  'auto' (this benchmark's choice; the trainer's own default is "
                         "'f32') = bf16 roundings in single-pass bf16 arithmetic, where every consumer multiplies by the bf16 values anyway: "
                         "the same step at half the bytes; LstmCtcTrainer.state_dict() exchanges the fp32 masters before a checkpoint")
    ap.add_argument('--grad-dtype', choices=['f32', 'bf16'], default='f32',
                    help='wire format of the data-parallel gradient all-reduce (N > 1)')
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 outside torchrun: run the N ranks as a child job.  Nothing in this process has initialised HIP (torch is not
    even imported yet), so no GPU state is inherited and no exec of a GPU-holding process happens."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.run(cmd, env=env)
    raise SystemExit(proc.returncode)


def param_count(hidden=H, layers=L):
    """Parameters of Encoder(80, 128, hidden) with `layers` LSTM layers + TemporalClassifier(hidden, 32): 13,207,712 for LC-2x1024."""
    lstm = sum(4 * hidden * ((C_SUB if l == 0 else hidden) + hidden) + 8 * hidden for l in range(layers))
    return C_SUB * F * 5 + C_SUB + lstm + V * hidden + V


def algorithmic_step_bytes(B, hidden=H, layers=L):
    """10 * 4 * P parameter-side bytes + per utterance the activations: x read (25,600), conv out written + read
    (2 x 10,752), the LSTM's saved gates / c / h (6H x 4 B x T' x L) written forward and read backward, logits / log-probs (3 x 2,688):
    2.12 MB per utterance for LC-2x1024."""
    per_utt = 2 * 6 * hidden * 4 * T_SUB * layers + 55_616
    return 10 * 4 * param_count(hidden, layers) + per_utt * B


MFMA_BF16_PEAK_TFS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA


def algorithmic_step_flops(B, hidden=H, layers=L):
    """forward = 2 x (conv 128 x 80 x 5 + per layer 4 hidden x (in + hidden) + classifier hidden x 32) MACs per frame
    x T' frames per utterance; a training step = 3 x forward (554 MFLOP -> 1.66 GFLOP per utterance for LC-2x1024)."""
    per_frame = C_SUB * F * 5 + sum(4 * hidden * ((C_SUB if l == 0 else hidden) + hidden) for l in range(layers)) + hidden * V
    return 3 * 2 * per_frame * T_SUB * B


def chain_algorithmic_bytes(B, direction, layers=1, strict=False):
    """Bytes ONE launch of the recurrent chain must move over its T' steps, SURVEY.md 8d accounting: every weight matrix the launch
    multiplies by once per pass (4 bytes per parameter), plus per step, utterance and layer the fp32 activations that enter or leave.
    forward : W_hh [4H,H]; per step: gate pre-activations in [4H], activated gates out [4H], c_t out [H], h_t out [H]
    backward: W_hh^T;      per step: activated gates in [4H], c_t in [H] (c_{t-1} is the same array), dh from above in [H],
                           gate gradients out [4H]
    layers = 2 (the two-layer launch, csrc/lstm_persist2.hip): both layers' terms plus W_ih of layer 1 [4H,H], which that launch
    multiplies by as well (forward: the input projection; backward: the input gradient)"""
    per_step = (4 * H + 4 * H + H + H) if direction == 'fwd' else (4 * H + H + H + 4 * H)
    if strict:
        per_step = 6 * H             # SURVEY.md 8d's own activation term: the saved gates / c / h, 6H per step, written forward, read backward
    weights = 1 if layers == 1 else 3
    return 4 * weights * (4 * H * H) + 4 * T_SUB * B * per_step * layers


def build_model(device, seed=42, hidden=H, layers=L):
    from haloop_amd import rnn, recognizer, synth
    enc_p, rec_p = synth.make_params(F, C_SUB, hidden, layers, V, seed)
    enc = rnn.Encoder(F, C_SUB, hidden, num_layers=layers)
    rec = recognizer.TemporalClassifier(hidden, V)
    enc.load_state_dict(enc_p)
    rec.load_state_dict(rec_p)
    return enc.to(device).train(), rec.to(device).train(), (enc_p, rec_p)


def time_chain(device, direction, layers, reps=20):
    """Median duration of the dominant kernel: the recurrent chain of the H=1024, B=64, T'=21 stack (layers = 2: the two-layer
    persistent launch; 1: one layer's chain).  The library records two HIP events on its launch stream right around the chain
    (halo_lstm_chain_events), so the batched GEMMs and operand preparation of the same call are outside the bracket.
    Returns (microseconds per chain, launches per chain, kernel name)."""
    import torch
    from haloop_amd import _lib, ops
    g = torch.Generator().manual_seed(0)
    in0 = C_SUB if layers == 2 else H
    x = (torch.randn(T_SUB, B_PER_GPU, in0, generator=g) * 0.1).to(device)
    w_ih = [(torch.rand(4 * H, in0 if l == 0 else H, generator=g) - 0.5).mul(0.06).to(device) for l in range(layers)]
    w_hh = [(torch.rand(4 * H, H, generator=g) - 0.5).mul(0.06).to(device) for l in range(layers)]
    b = [torch.zeros(4 * H, device=device) for l in range(layers)]
    dy = (torch.randn(B_PER_GPU, T_SUB, H, generator=g) * 0.01).to(device)
    drop = ops.Dropout(0.2, 1, 0) if layers == 2 else ops.NO_DROPOUT
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); e1.record()
    torch.cuda.synchronize()
    grads = {'dw_ih': [torch.zeros_like(w) for w in w_ih], 'dw_hh': [torch.zeros_like(w) for w in w_hh],
             'db_ih': [torch.zeros_like(v) for v in b], 'db_hh': [torch.zeros_like(v) for v in b]}
    ts = []
    for i in range(reps + 3):
        if direction == 'fwd':
            _lib.lstm_chain_events(e0, e1)
        y, _, _, reserve = ops.lstm_fwd(x, w_ih, w_hh, b, b, y_strides=None, y_relu=False, drop=drop)
        if direction == 'bwd':
            _lib.lstm_chain_events(e0, e1)
            ops.lstm_bwd(x, w_ih, w_hh, dy, (H, T_SUB * H), False, reserve, grads=grads, drop=drop)
        _lib.lstm_chain_events(None, None)
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
    info = _lib.lstm_chain_info(direction)
    ts.sort()
    return 1e3 * ts[len(ts) // 2], info['launches'], info['kernel']          # median: one preempted sample must not move it


def time_inference(enc, rec, x, steps):
    """Second half of the metric: forward-only + greedy decode, eval mode."""
    import torch
    from haloop_amd.infer import LstmCtcRecognizer
    was_training = enc.training

    def run(use_graph, n):
        reco = LstmCtcRecognizer(enc, rec, use_graph=use_graph)
        for _ in range(5):
            reco.recognize(x)
        xs = reco.static_input() if reco.static_input() is not None else x      # resident in the graph's own buffer
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            reco.recognize(xs, clone=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n
    # six launches per batch: replayed from a HIP graph or launched eagerly, whichever is faster on this box (as the training step)
    probe = {True: [], False: []}
    for _ in range(3):                                  # alternating rounds; eager only if its worst round beats the best replayed one
        probe[True].append(run(True, 30))
        probe[False].append(run(False, 30))
    use_graph = not (max(probe[False]) < min(probe[True]))
    dt = run(use_graph, steps)
    enc.train(was_training); rec.train(was_training)
    return {'metric': 'utterances/sec, forward + greedy CTC decode (eval)', 'value': round(x.shape[0] / dt, 1),
            'unit': 'utterances/s', 'ms_per_batch': round(1e3 * dt, 4), 'batch': x.shape[0], 'hip_graph': use_graph,
            'launch_mode_probe': {'graph_replay_ms': [round(1e3 * t, 4) for t in probe[True]], 'eager_launches_ms': [round(1e3 * t, 4) for t in probe[False]]}}


def time_other_mode(mode, device, batch, warmup, steps, use_graph, hidden=H, layers=L):
    """The same training step in another arithmetic mode, at another batch size, or on another of the reference's model shapes
    (own model + trainer, same seeds)."""
    import torch
    from haloop_amd import _lib
    from haloop_amd.train import LstmCtcTrainer
    _lib.set_math_mode(mode)
    enc, rec, _ = build_model(device, hidden=hidden, layers=layers)
    tr = LstmCtcTrainer(enc, rec, seed=1337, use_graph=True, alias_loss=True)
    for _ in range(warmup):
        tr.step(*batch)
    if use_graph is None:           # graph replay or eager launches, whichever is faster here (as the headline)
        def probe(n):
            for _ in range(3):
                tr.step(*batch)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                tr.step(*batch)
            torch.cuda.synchronize()
            return time.perf_counter() - t
        g_t, e_t = [], []
        for _ in range(3):                          # eager only if its worst round beats the best replayed one
            tr.use_graph = True
            g_t.append(probe(10))
            tr.use_graph = False
            e_t.append(probe(10))
        use_graph = not (max(e_t) < min(g_t))
    tr.use_graph = bool(use_graph)
    if not tr.use_graph:
        for _ in range(3):
            tr.step(*batch)
    b2 = (tr.static_inputs() or batch) if tr.use_graph else batch
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        tr.step(*b2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    tr.check_status()
    B = batch[0].shape[0]
    nbytes = algorithmic_step_bytes(B, hidden, layers)
    flops = algorithmic_step_flops(B, hidden, layers)
    return {'value': round(B * steps / dt, 1), 'unit': 'utterances/s', 'ms_per_step': round(1e3 * dt / steps, 4), 'batch': B,
            'dtype': MATH_DTYPE[mode], 'final_loss': round(tr.loss.item(), 5), 'hip_graph': tr.use_graph,
            'step_frac_of_hbm_peak': round(nbytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
            # the same step against the dense bf16 MFMA roof (forward 554 MFLOP per utterance at LC-2x1024, a training step 3x):
            # the bound that takes over from HBM as the batch grows (bytes(B) = 528 MB + 2.1 MB B, flops(B) = 1.66 GFLOP B)
            'step_frac_of_mfma_peak': round(flops / (dt / steps) / 1e12 / MFMA_BF16_PEAK_TFS, 4),
            'bound_us': {'hbm': round(nbytes / (HBM_PEAK_GBS * 1e9) * 1e6, 1), 'mfma': round(flops / (MFMA_BF16_PEAK_TFS * 1e12) * 1e6, 1)}}


def batch_sweep(mode, device, warmup, steps, use_graph):
    """'larger B raises the achieved fraction.  Report both.'  The same step at B = 128, 256 and 512 per GPU
    (bytes(B) = 10*4*P + 2.12e6*B, flops(B) = 1.66e9*B): against the HBM roof and against the MFMA roof, which takes over as the
    batch grows; in bf16 arithmetic the two-layer persistent launches take two batch tiles per workgroup."""
    from haloop_amd import _lib, synth
    out = {}
    for B in (128, 256, 512):
        batch = tuple(t.to(device) for t in synth.synthetic_batch(B, T, F, V, S, 42))
        r = time_other_mode(mode, device, batch, warmup, steps, use_graph)
        r['recurrence'] = _lib.lstm_chain_info('bwd')['kernel']
        out[f'B{B}'] = r
    return out


def run_config_leg(script, mode, timeout_s):
    """BASELINE configs 3 / 5: the tool's own process (its JSON line is its last line of output), after this process's timed
    region; a failure or a timeout is recorded, never fatal to the headline."""
    env = dict(os.environ, HALO_MATH=mode)
    try:
        proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script)], env=env, capture_output=True, text=True,
                              timeout=timeout_s)
    except subprocess.TimeoutExpired:
        return {'error': f'{script} exceeded {timeout_s} s'}
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
    if proc.returncode != 0 or not lines:
        return {'error': f'{script} exit code {proc.returncode}', 'stderr_tail': proc.stderr[-400:]}
    return json.loads(lines[-1])


def host_cores():
    """Cores this process may actually use (affinity mask and cgroup quota), not the machine's count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get('HALO_CPU_THREADS', '16'))))


def _cpu_leg(params, B, budget_s, max_steps):
    import torch
    from oracle import cpu_ref
    enc_p, rec_p = params
    tr = cpu_ref.Trainer(enc_p, rec_p)
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F, V, S, 42)
    for _ in range(2):
        tr.step(x, il, tg, tl, p_torch=0.2)
    n, t0 = 0, time.perf_counter()
    while True:
        tr.step(x, il, tg, tl, p_torch=0.2)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= max_steps:
            break
    return n, dt


def cpu_baseline(params):
    """The oracle's Trainer (stock torch CPU ops = what the reference runs on CPU, ha/loop.py:176-196) on this box's host cores:
    the bench workload (B=64) and BASELINE config 1 (B=4)."""
    import torch
    torch.set_num_threads(host_cores())
    n64, dt64 = _cpu_leg(params, B_PER_GPU, 16.0, 200)
    n4, dt4 = _cpu_leg(params, 4, 8.0, 400)
    return {'value': round(n64 * B_PER_GPU / dt64, 2), 'unit': 'utterances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n64} training steps of the same B={B_PER_GPU} workload (stock torch CPU ops: conv1d, nn.LSTM kernel, '
                      f'ctc_loss, clip, AdamW), {dt64:.1f} s',
            'config1_b4': {'value': round(n4 * 4 / dt4, 2), 'unit': 'utterances/s', 'ms_per_step': round(1e3 * dt4 / n4, 2),
                           'sample': f'{n4} training steps at B=4 (BASELINE.json configs[0]), {dt4:.1f} s'}}


def read_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (tools/pmc_traffic.py writes the
    file from the FETCH_SIZE / WRITE_SIZE CSVs with the guide's gfx950 corrections); null unless it is for THIS kernel.  The newest
    round's file first (profiles/rNN_lstm2_chain_traffic.json), the one-layer-per-launch file of round 2 last."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_lstm2_chain_traffic.json')), reverse=True)
    names.append(os.path.join(ROOT, 'profiles', 'lstm_chain_traffic.json'))
    for tpath in names:
        if not os.path.exists(tpath):
            continue
        for rec in json.load(open(tpath)).get('kernels', []):
            if rec.get('kernel') and (rec['kernel'] in kernel_name or kernel_name in rec['kernel']):
                return rec.get('hbm_bytes_per_launch')
    return None


TRACE_US = {}        # kernel name -> median duration (us) in the rocprofv3 kernel trace of measure_traffic's first pass


def trace_us(name):
    for k, v in TRACE_US.items():
        if name in k:
            return round(v, 3)
    return None


def measure_traffic(math):
    """HBM bytes per launch of the recurrent chains, measured IN THIS RUN: two child processes under rocprofv3, one PMC counter per
    pass (FETCH_SIZE, then WRITE_SIZE) as MI355X_MICROARCH.md's HBM section prescribes, on tools/run_lstm2_steps.py (the benchmark's
    LSTM stack alone: same kernels, same shapes); bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (that section's gfx950 correction
    for 16-byte-per-lane reads).  Returns {kernel name: bytes} or None when rocprofv3 is not available or a pass fails (the
    committed passes under profiles/ are used then)."""
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3')
    if not exe:
        return None
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import pmc_traffic
    work = tempfile.mkdtemp(prefix='halo_pmc_', dir='/tmp')
    env = dict(os.environ, HALO_MATH=math, TMPDIR='/tmp')
    found = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(work, counter)
            proc = subprocess.run([exe, '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable,
                                   os.path.join(ROOT, 'tools', 'run_lstm2_steps.py')], env=env, cwd='/tmp', capture_output=True, text=True,
                                  timeout=180)
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('counter_collection.csv')]
            if proc.returncode != 0 or not csvs:
                return None
            found[counter] = pmc_traffic.medians(csvs[0], counter)
            if counter == 'FETCH_SIZE':          # the same pass's kernel trace: begin / end time stamps of every launch
                import csv as _csv
                import statistics
                per = {}
                for tr in (os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('kernel_trace.csv')):
                    for r in _csv.DictReader(open(tr)):
                        per.setdefault(r['Kernel_Name'], []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
                TRACE_US.update({k: statistics.median(v) for k, v in per.items()})
    except (subprocess.TimeoutExpired, OSError):
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    res = {}
    for k in set(found['FETCH_SIZE']) | set(found['WRITE_SIZE']):
        res[k] = int(round(2 * found['FETCH_SIZE'].get(k, (0.0, 0))[0] * 1024 + found['WRITE_SIZE'].get(k, (0.0, 0))[0] * 1024))
    return res


def main():
    args = parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    device = torch.device('cuda', 0 if args.share_gpu else local_rank)
    torch.cuda.set_device(device)
    if world > 1 and args.share_gpu:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('gloo')
    elif world > 1:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=device)
    elif args.dp_rehearsal:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
        if args.dp_algo == 'allreduce':
            from haloop_amd import dp
            dp.world_size = lambda group=None: 2      # every world > 1 branch; an all-reduce (AVG) over one rank is the identity

    from haloop_amd import _lib, synth
    from haloop_amd.train import LstmCtcTrainer
    _lib.lib()                                              # loud failure if the HIP library is absent
    _lib.set_math_mode(args.math)
    if args.share_gpu and world > 1:
        # two processes on one GPU: the persistent recurrences need the whole chip to themselves (their bounded waits would fail the
        # step, loudly); the rehearsal runs the step-launch chain in front of the data-parallel tail it is about
        _lib.set_lstm_persistent(False)

    enc, rec, params = build_model(device)
    dp_kw = dict(dp_algo=args.dp_algo, rehearse_dp=args.dp_rehearsal and args.dp_algo != 'allreduce', gather_dtype=args.gather_dtype)
    trainer = LstmCtcTrainer(enc, rec, seed=1337 + rank, use_graph=not args.no_graph, grad_dtype=args.grad_dtype, alias_loss=True, **dp_kw)
    x, il, tg, tl = (t.to(device) for t in synth.synthetic_batch(B_PER_GPU, T, F, V, S, 42 + rank))

    use_graph = not args.no_graph
    try:
        for _ in range(args.warmup):
            trainer.step(x, il, tg, tl)
        torch.cuda.synchronize()
    except Exception as e:              # e.g. a collective backend that cannot live beside stream capture: run the step eagerly
        if not use_graph:
            raise
        print(f'[bench rank {rank}] graph mode failed ({type(e).__name__}: {e}); falling back to eager launches', file=sys.stderr, flush=True)
        use_graph = False
        enc, rec, params = build_model(device)
        trainer = LstmCtcTrainer(enc, rec, seed=1337 + rank, use_graph=False, grad_dtype=args.grad_dtype, alias_loss=True, **dp_kw)
        for _ in range(args.warmup):
            trainer.step(x, il, tg, tl)
    mode_probe = None
    if world == 1 and not args.dp_rehearsal and use_graph and not args.graph and not args.no_graph:
        # graph replay or eager launches: whichever is faster on this box (both are the same launches on the same stream)
        def probe(n):
            for _ in range(5):
                trainer.step(x, il, tg, tl)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                trainer.step(x, il, tg, tl)
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t) / n
        # three alternating rounds of 30 steps; eager launches depend on the host keeping ahead, so they are chosen only if their WORST
        # round beats the graph's BEST (a busy host shows up as scatter between the eager rounds)
        g_ms, e_ms = [], []
        for _ in range(3):
            trainer.use_graph = True
            g_ms.append(probe(30))
            trainer.use_graph = False
            e_ms.append(probe(30))
        use_graph = not (max(e_ms) < min(g_ms))
        trainer.use_graph = use_graph
        mode_probe = {'graph_replay_ms': [round(t, 4) for t in g_ms], 'eager_launches_ms': [round(t, 4) for t in e_ms], 'steps_each': 30,
                      'rule': 'eager iff its worst round beats the best replayed round'}
    if use_graph and trainer.static_inputs() is not None:   # inputs resident in the step graph's own buffers (no per-step copy)
        x, il, tg, tl = trainer.static_inputs()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(x, il, tg, tl)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_ranks_seen = 1
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)                                # every rank that really took part adds one
        n_ranks_seen = int(ones.item())
    loss = trainer.loss.item()
    trainer.check_status()        # a persistent recurrence that gave up a wait during the run: fail the bench, loudly
    dp_components = None
    if (world > 1 or args.dp_rehearsal) and trainer.sharded is not None:
        # after the timed region: the same step launched eagerly with a HIP event behind every collective and optimizer piece (rank 0's
        # view; every rank takes part in the collectives)
        dp_components = trainer.profile_dp_components(x, il, tg, tl, steps=5)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = world * B_PER_GPU * args.steps / elapsed
        step_bytes = algorithmic_step_bytes(B_PER_GPU)
        out = {
            'metric': 'utterances/sec, LSTM-CTC training step (fwd + CTC loss + bwd + clip + AdamW)',
            'value': round(value, 1), 'unit': 'utterances/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': MATH_DTYPE[args.math], 'data': 'synthetic',
            'config': {'workload': 'LC-2x1024: conv(80->128,k5,s4) + 2-layer LSTM H=1024 + Linear(1024->32) + CTC, '
                                   '80 frames x 80 mels, vocab 32, targets 5-10 symbols, dropout 0.2',
                       'batch_per_gpu': B_PER_GPU, 'global_batch': world * B_PER_GPU, 'frames': T, 'mels': F,
                       'parallelism': f'dp{world}' + (' (data-parallel code path rehearsed on one rank)' if args.dp_rehearsal else '')
                                      + (' (REHEARSAL: all ranks share ONE GPU, gloo rendezvous -- not a scaling number)' if args.share_gpu and world > 1 else ''), 'hip_graph': use_graph, 'launch_mode_probe': mode_probe, 'math': args.math,
                       'grad_allreduce_dtype': args.grad_dtype if world > 1 else None,
                       'dp_algo': trainer.dp_algo if (world > 1 or args.dp_rehearsal) else None,
                       'dp_gather_dtype': (('bf16' if getattr(trainer.sharded, 'gather_bf16', False) else 'f32') if (world > 1 or args.dp_rehearsal) else None),
                       'dp_collectives_captured': getattr(trainer, '_tail_graph', None) is not None if (world > 1 or args.dp_rehearsal) else None,
                       'dp_early_reduce_scatter_overlapped': bool(getattr(trainer, '_early_started', False)) if (world > 1 or args.dp_rehearsal) else None,
                       'dp_wire_bytes_per_rank': (trainer.sharded.wire_bytes() if hasattr(getattr(trainer, 'sharded', None), 'wire_bytes') else None),
                       'dp_components_us': dp_components,
                       'dp_launch_modes': ({'forward_backward': 'eager launches' if (not use_graph or trainer.eager_forward_backward) else 'graph replay',
                                            'tail (collectives + optimizer)': 'graph replay' if getattr(trainer, '_tail_graph', None) is not None else 'eager launches',
                                            'note': 'the N = 1 line of a scaling curve runs the plain step (graph replay or eager launches, whichever its probe '
                                                    'chose: config.hip_graph / launch_mode_probe); N > 1 skips the probe'}
                                           if (world > 1 or args.dp_rehearsal) else None)},
            'n_ranks_seen': n_ranks_seen,
            'final_loss': round(loss, 5), 'steps_trained': args.warmup + args.steps + (210 if mode_probe else 0),
            'step_roofline': {'algorithmic_bytes_per_step': step_bytes,
                              'achieved_GBs': round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
                              'frac_of_hbm_peak': round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              'note': 'whole training step against SURVEY.md 8d bytes(B) = 10*4*P + 2.12e6*B'},
        }
        if world == 1:
            # dominant kernel: the backward recurrent chain (bf16 mode: ONE launch for both layers, otherwise one per layer); the
            # forward twin beside it
            two = bool(_lib.lib().halo_lstm_persistent2_eligible(T_SUB, B_PER_GPU, H, L))
            cl = 2 if two else 1
            us_b, launches_b, name_b = time_chain(device, 'bwd', cl)
            us_f, launches_f, name_f = time_chain(device, 'fwd', cl)
            kb, kf = chain_algorithmic_bytes(B_PER_GPU, 'bwd', cl), chain_algorithmic_bytes(B_PER_GPU, 'fwd', cl)
            chains = L // cl
            live = None if args.no_extras else measure_traffic(args.math)

            def traffic(name):
                if live:
                    for k, v in live.items():
                        if name in k:
                            return v
                return read_traffic(name)
            # `frac` = SURVEY.md 8d's own byte count (weights once + the 6H saved values per step, layer and utterance); `frac_10h` keeps
            # the rounds 1-4 yardstick (the 10H fp32 values a per-layer chain reads and writes; the two-layer launch no longer stores
            # the 4H fp32 gate gradients that yardstick credits it with)
            kb_s, kf_s = chain_algorithmic_bytes(B_PER_GPU, 'bwd', cl, strict=True), chain_algorithmic_bytes(B_PER_GPU, 'fwd', cl, strict=True)
            ach = kb_s / launches_b / (us_b / launches_b * 1e-6) / 1e9
            out['roofline'] = {
                'bound': 'hbm', 'kernel': name_b, 'launches_per_chain': launches_b, 'chains_per_step': chains, 'layers_per_chain': cl,
                'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                'frac_10h': round(kb / (us_b * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                'algorithmic_bytes_per_launch_10h': kb // launches_b,
                'traffic': traffic(name_b),
                'traffic_source': ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run by this process (tools/run_lstm2_steps.py), bytes = 2 F + W'
                                   if live else 'committed passes under profiles/ (rocprofv3 not available to this run)'),
                'algorithmic_bytes_per_launch': kb_s // launches_b, 'avg_launch_us': round(us_b / launches_b, 3),
                # the same launch in the kernel trace of this run's FETCH_SIZE counter pass (median over its launches; under the counters
                # a launch runs ~5 % slower than in a plain `rocprofv3 --kernel-trace --stats`, whose averages are committed under
                # profiles/): the HIP-event bracket lies between the two
                'avg_launch_us_kernel_trace': trace_us(name_b),
                'accounting': 'SURVEY.md 8d: every weight matrix the launch multiplies by once per pass (W_hh^T per layer; the two-layer '
                              'launch also W_ih of layer 1) at 4 B per parameter + per step, layer and utterance the 6H saved fp32 values '
                              '(gates, c, h), divided over the launches of the chain',
                'share_of_step': round(chains * us_b * 1e-3 / ms_per_step, 3),
                'forward_twin': {'kernel': name_f, 'launches_per_chain': launches_f, 'avg_launch_us': round(us_f / launches_f, 3),
                                 'avg_launch_us_kernel_trace': trace_us(name_f),
                                 'algorithmic_bytes_per_launch': kf_s // launches_f,
                                 'achieved': round(kf_s / (us_f * 1e-6) / 1e9, 1),
                                 'frac': round(kf_s / (us_f * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                 'frac_10h': round(kf / (us_f * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), 'traffic': traffic(name_f),
                                 'share_of_step': round(chains * us_f * 1e-3 / ms_per_step, 3)}}
        if world == 1 and not args.no_extras:
            out['inference'] = time_inference(enc, rec, x, max(20, args.steps // 2))
            out['inference']['dtype'] = MATH_DTYPE[args.math]
            fwd_bytes = 4 * param_count() + 28_300 * B_PER_GPU          # SURVEY.md 8d, forward only: 4P + 28.3e3 B = 54.6 MB at B = 64
            out['inference']['roofline'] = {'bound': 'hbm', 'algorithmic_bytes_per_batch': fwd_bytes, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                            'achieved': round(fwd_bytes / (out['inference']['ms_per_batch'] * 1e-3) / 1e9, 1),
                                            'frac': round(fwd_bytes / (out['inference']['ms_per_batch'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if args.math != 'bf16x3':
                # the same leg in the fp32-grade arithmetic, the mode whose greedy alignments are exact on the reference's random-init
                # fixtures (bf16's are exact on peaked posteriors: tests/test_gpu_bf16_contract.py)
                _lib.set_math_mode('bf16x3')
                out['inference_bf16x3'] = time_inference(enc, rec, x, max(20, args.steps // 2))
                out['inference_bf16x3']['dtype'] = MATH_DTYPE['bf16x3']
                _lib.set_math_mode(args.math)
            n2 = max(20, args.steps // 2)
            launch_mode = False if args.no_graph else (True if args.graph else None)
            for mode in ('bf16', 'bf16x3', 'f32'):
                if mode != args.math:
                    out[mode + '_mode'] = time_other_mode(mode, device, (x, il, tg, tl), args.warmup, n2, launch_mode)
            # the fp32-grade step in the parsed headline's config, not only in an extra key
            out['config']['bf16x3_ms_per_step'] = out.get('bf16x3_mode', out)['ms_per_step']
            out['config']['bf16x3_utterances_per_s'] = out.get('bf16x3_mode', out)['value']
            # the reference's own model shapes: the stock 3-layer encoder (ha/rnn.py:6-11; layer 0 one persistent
            # launch, the top pair one two-layer launch) and the H = 1536 variant (ha/init.py:171; per-layer 32-row launches)
            out['stock3'] = time_other_mode(args.math, device, (x, il, tg, tl), args.warmup, n2, launch_mode, hidden=1024, layers=3)
            out['stock3'].update(model='Encoder(80,128,1024), stock 3-layer nn.LSTM (ha/rnn.py:6-11) + TemporalClassifier(1024,32)',
                                 params=param_count(1024, 3), recurrence=_lib.lstm_chain_info('bwd')['kernel'])
            out['H1536'] = time_other_mode(args.math, device, (x, il, tg, tl), args.warmup, n2, launch_mode, hidden=1536, layers=2)
            out['H1536'].update(model='Encoder(80,128,1536), 2-layer LSTM (ha/init.py:171 width) + TemporalClassifier(1536,32)',
                                params=param_count(1536, 2), recurrence=_lib.lstm_chain_info('bwd')['kernel'])
            out['b_sweep'] = batch_sweep(args.math, device, args.warmup, max(20, args.steps // 4), use_graph)       # (the launch mode the headline chose)
            out['b_sweep'][f'B{B_PER_GPU}'] = {'value': out['value'], 'unit': 'utterances/s', 'ms_per_step': out['ms_per_step'], 'batch': B_PER_GPU,
                                               'step_frac_of_hbm_peak': out['step_roofline']['frac_of_hbm_peak'],
                                               'recurrence': out.get('roofline', {}).get('kernel')}
            _lib.set_math_mode(args.math)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(params)
        if world == 1 and not args.no_extras and not args.no_configs:
            # BASELINE configs 3 and 5, each in its own process after everything of this one is timed
            torch.cuda.synchronize()
            out['gpt2_small'] = run_config_leg('bench_gpt.py', 'bf16', 240)
            out['asr_transformer32'] = run_config_leg('bench_asr.py', 'bf16x3', 240)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
