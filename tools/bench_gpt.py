#!/usr/bin/env python3
"""BASELINE config 3: GPT-2 small (124 M), seq_len 1024, on one MI355X -- tokens/s of scoring (`hap`: forward_all,
reduction='none') and of the full training step (`hal`: forward_all + loss.backward() + AdamW on the HIP kernels),
nats/token against the CPU oracle, the dominant kernel against the MFMA roof, and the CPU oracle timed on the host
cores.  Prints human-readable lines and ONE JSON line (last).

    HALO_MATH=bf16x3|bf16|f32  B=8  python tools/bench_gpt.py [--no-cpu-baseline]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from haloop_amd import _lib, attention, ops, synth

MFMA_BF16_PEAK = 2500.0     # TFLOP/s dense bf16, MI355X_MICROARCH.md
_lib.lib(); _lib.lend_scratch(256 << 20)
math_mode = os.environ.get('HALO_MATH', 'bf16x3'); _lib.set_math_mode(math_mode)
B = int(os.environ.get('B', '8')); T = 1024
cfg = attention.GPTConfig()
torch.manual_seed(0)
model = attention.GPT(cfg).cuda().eval()
with torch.no_grad():                                     # the reference's init zeroes wpe; give it content
    model.transformer.wpe.weight.normal_(0, 0.02)
inputs, targets = synth.synthetic_tokens(B, T, cfg.vocab_size, 3, pad_tail=False)
inputs_d, targets_d = inputs.cuda(), targets.cuda()


def timed(fn, n=5, warm=2):
    for _ in range(warm): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n


def event_us(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


with torch.inference_mode():
    per_tok, t_fwd = timed(lambda: model.forward_all(inputs_d, targets_d, reduction='none'))
print(f'GPT-2 small forward_all B={B} T={T} math={math_mode}: {t_fwd*1e3:.2f} ms  {B*T/t_fwd:,.0f} tokens/s  '
      f'mean nats/token {per_tok.mean().item():.4f}')

model.train()
params = list(model.parameters())
# one fused AdamW launch over the whole parameter list (the reference trains with torch.optim.AdamW(fused=True))
optimizer = ops.AdamWMulti(params, [0.1 if p.dim() >= 2 else 0.0 for p in params], lr=3e-4, betas=(0.9, 0.95), eps=1e-8)


def train_step():
    for p in params: p.grad = None
    loss = model.forward_all(inputs_d, targets_d)
    loss.backward()
    optimizer.step()
    return loss


loss, t_train = timed(train_step)
print(f'GPT-2 small train step (fwd+bwd+AdamW) B={B} T={T} math={math_mode}: {t_train*1e3:.2f} ms  {B*T/t_train:,.0f} tokens/s  '
      f'loss {loss.item():.4f}')

# algorithmic FLOPs: 2 * P_nonemb per token, causal attention 4*T^2*d*L/2 per sequence, lm_head 2*d*V per token
C, L, V = cfg.n_embd, cfg.n_layer, cfg.vocab_size
p_nonemb = sum(p.numel() for n, p in model.named_parameters() if 'wte' not in n and 'wpe' not in n and 'lm_head' not in n)
fwd_flops = B * (2 * p_nonemb * T + 4 * T * T * C * L // 2 + 2 * C * V * T)
train_flops = 3 * fwd_flops
# the path's product kernel on the MLP up-projection shape [B*T, 768] x [3072, 768]^T, as the step launches it (bf16 arithmetic: halo_gemm_rows,
# row-major bf16 activations -> bf16 result; otherwise the split GEMM on operand images), timed with HIP events on the launch stream
M, N, K = B * T, 4 * C, C
b_img = ops.split_image(torch.randn(N, K, device='cuda'))
rows_kernel = ops.gemm_rows_supported(M, N, K)
if rows_kernel:
    a_b = torch.randn(M, K, device='cuda').bfloat16()
    gemm_us = event_us(lambda: ops.gemm_rows(a_b, b_img, M, N, K, out_bf16=True))
    kname = 'gemm_rows_kernel<6, 0, false> (256 x 192 tiles, 1 bf16 MFMA pass, bf16 result)'
else:
    a_img = ops.split_image(torch.randn(M, K, device='cuda'))
    out = torch.empty(M, N, device='cuda')
    gemm_us = event_us(lambda: ops.gemm_split(a_img, b_img, M, N, K, out=out))
passes = 1 if math_mode == 'bf16' else 3
if not rows_kernel:
    kname = f'gemm_bf16x3_kernel ({passes} bf16 MFMA pass(es))'
gemm_tflops = passes * 2.0 * M * N * K / (gemm_us * 1e-6) / 1e12


def pmc_traffic():
    """HBM bytes per launch of that product: two rocprofv3 --pmc child passes on tools/pmc_gemm.py (FETCH_SIZE, then WRITE_SIZE), bytes =
    2 F + W with the guide's gfx950 correction (MI355X_MICROARCH.md, HBM).  None when rocprofv3 is not there or a pass fails."""
    import shutil, subprocess, tempfile
    exe = shutil.which('rocprofv3')
    if not exe or '--no-pmc' in sys.argv:
        return None
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import pmc_traffic as pt
    work = tempfile.mkdtemp(prefix='halo_pmc_gpt_', dir='/tmp')
    vals = {}
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = os.path.join(work, counter)
            proc = subprocess.run([exe, '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable,
                                   os.path.join(os.path.dirname(os.path.abspath(__file__)), 'pmc_gemm.py'), str(M), str(N), str(K)],
                                  env=dict(os.environ, HALO_MATH=math_mode, TMPDIR='/tmp'), cwd='/tmp', capture_output=True, text=True, timeout=150)
            csvs = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith('counter_collection.csv')]
            if proc.returncode != 0 or not csvs:
                return None
            med = pt.medians(csvs[0], counter)
            hit = [v for k, v in med.items() if ('gemm_rows_kernel' in k if rows_kernel else 'gemm_bf16x3_kernel' in k)]
            if not hit:
                return None
            vals[counter] = hit[0][0]
    except (subprocess.TimeoutExpired, OSError):
        return None
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return int(round(2 * vals['FETCH_SIZE'] * 1024 + vals['WRITE_SIZE'] * 1024))


res = {
    'metric': 'tokens/sec, GPT-2 small LM seq_len 1024 (BASELINE config 3): train step (fwd + bwd + AdamW) and scoring',
    'value': round(B * T / t_train, 1), 'unit': 'tokens/s', 'n_gpus': 1, 'ms_per_step': round(t_train * 1e3, 3),
    'scoring': {'value': round(B * T / t_fwd, 1), 'unit': 'tokens/s', 'ms_per_batch': round(t_fwd * 1e3, 3)},
    'dtype': {'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMAs per product, fp32 accumulate)', 'bf16': 'bf16', 'f32': 'f32'}[math_mode], 'data': 'synthetic',
    'config': {'workload': 'GPT-2 small 12L/12h/768, vocab 50304, tied lm_head, random init', 'batch': B, 'seq_len': T, 'math': math_mode},
    'step_mfma': {'algorithmic_tflop_per_train_step': round(train_flops / 1e12, 3),
                  'achieved_tflops': round(train_flops / t_train / 1e12, 1),
                  'mfma_tflops_issued': round(passes * train_flops / t_train / 1e12, 1), 'peak': MFMA_BF16_PEAK},
    'roofline': {'bound': 'mfma', 'kernel': f'{kname}, M={M} N={N} K={K}',
                 'achieved': round(gemm_tflops, 1), 'peak': MFMA_BF16_PEAK, 'unit': 'TFLOP/s', 'frac': round(gemm_tflops / MFMA_BF16_PEAK, 4),
                 'traffic': None, 'algorithmic_bytes_per_launch': 2 * (M * K + N * K + M * N), 'avg_launch_us': round(gemm_us, 1)},
}

torch.cuda.synchronize()
res['roofline']['traffic'] = pmc_traffic()
res['roofline']['traffic_source'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (tools/pmc_gemm.py), bytes = 2 F + W' if res['roofline']['traffic'] else None
if '--no-cpu-baseline' not in sys.argv:
    from oracle import gpt_ref                            # the checker and the CPU baseline: nothing above touches oracle/
    # nats/token parity and the CPU baseline on ONE sequence (the oracle = stock torch CPU ops = what the reference runs on CPU)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cpu_p = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        t0 = time.perf_counter()
        ref_tok = gpt_ref.gpt_forward_all(cpu_p, L, cfg.n_head, inputs[:1], targets[:1], reduction='none')
        t_cpu_fwd = time.perf_counter() - t0
    model.eval()
    with torch.inference_mode():
        gpu_tok = model.forward_all(inputs_d[:1], targets_d[:1], reduction='none').cpu()
    err = (gpu_tok - ref_tok).abs()
    cpu_r = {k: v.requires_grad_(True) for k, v in cpu_p.items()}
    cpu_r['lm_head.weight'] = cpu_r['transformer.wte.weight']
    t0 = time.perf_counter()
    gpt_ref.gpt_forward_all(cpu_r, L, cfg.n_head, inputs[:1], targets[:1]).backward()
    t_cpu_train = time.perf_counter() - t0
    res['nats_per_token'] = {'hip_mean': round(gpu_tok.mean().item(), 5), 'cpu_mean': round(ref_tok.mean().item(), 5),
                             'max_abs_diff': round(err.max().item(), 6), 'mean_abs_diff': round(err.mean().item(), 7)}
    res['cpu_baseline'] = {'value': round(T / t_cpu_train, 1), 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                           'sample': f'one 1024-token sequence: forward {t_cpu_fwd:.2f} s, forward+backward {t_cpu_train:.2f} s (no optimizer)',
                           'scoring_tokens_per_s': round(T / t_cpu_fwd, 1)}
    print(f"nats/token HIP {res['nats_per_token']['hip_mean']} vs CPU {res['nats_per_token']['cpu_mean']} "
          f"(max abs diff {res['nats_per_token']['max_abs_diff']}); CPU {res['cpu_baseline']['value']} tokens/s train, "
          f"{res['cpu_baseline']['scoring_tokens_per_s']} tokens/s scoring on {res['cpu_baseline']['cores']} threads")
print(json.dumps(res), flush=True)
