#!/usr/bin/env python3
"""BASELINE config 5 shape: encoder-decoder attention ASR (`transformer:32`: 12+12 layers, 8x64 heads; ha/init.py:234-239)
on 80-frame x 80-mel utterances: encoder forward, CTC-head beam decode (beam 16, ha.beam semantics), attention greedy
decode (fp16 KV caches).  Prints utterances/s per stage and the WER of the HIP hypotheses against the CPU oracle's
(own edit distance) on a small sample, with the oracle timed on the host cores.  The oracle (oracle/transformer_ref.py) is used
here as the checker, as the CPU baseline and for its seeded parameter / batch generators; none of the timed HIP stages calls it."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from haloop_amd import _lib, transformer, beam
from oracle import transformer_ref as ref, lattice

_lib.lib(); _lib.lend_scratch(256 << 20)
math_mode = os.environ.get('HALO_MATH', 'bf16x3'); _lib.set_math_mode(math_mode)
N = int(os.environ.get('B', '64')); NCPU = int(os.environ.get('BCPU', '4')); V, HD, H, L = 32, 64, 8, 12
pe = ref.make_encoder_params(HD, H, L, 80, 256, 3, 28)
pd = ref.make_decoder_params(V, HD, H, L, 29)
enc = transformer.AudioEncoder(head_dim=HD, heads=H, layers=L, p_drop=0.2)
dec = transformer.CTCAttentionDecoder(vocab=V, head_dim=HD, heads=H, p_drop=0.2, layers=L)
enc.load_state_dict(pe); dec.load_state_dict(pd)
enc.cuda().eval(); dec.cuda().eval()
x, il, tg, tl = ref.synthetic_asr_batch(N, 80, 80, V, 8, 30, ragged=False)
xd, ild, tld = x.cuda(), il.cuda(), tl.cuda()


def edit_distance(a, b):
    d = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        prev, d[0] = d[0], i
        for j, cb in enumerate(b, 1):
            prev, d[j] = d[j], min(d[j] + 1, d[j - 1] + 1, prev + (ca != cb))
    return d[-1]


def timed(fn, n=5, warm=2):
    for _ in range(warm): out = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) / n


# training direction (`hala`): encoder -> decoder CE + 0.3 CTC -> backward, dropout 0.2 on, every parameter's gradient
enc.train(); dec.train()
cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1).cuda()
def train_fwd_bwd():
    for p in list(enc.parameters()) + list(dec.parameters()): p.grad = None
    f, fl, _ = enc(xd, ild)
    loss, _ = dec(f, cond, fl, (tl + 1).cuda())
    loss.backward()
    return loss
loss, t_train = timed(train_fwd_bwd)
print(f'transformer:32 N={N} joint-loss forward+backward (dropout 0.2), eager launches: {t_train*1e3:.2f} ms ({N/t_train:,.0f} utt/s), loss {loss.item():.4f}')
# the same forward + backward replayed from one HIP graph (haloop_amd.train.GraphedTrainStep; fresh dropout masks every replay)
from haloop_amd import train as _train
tl1 = (tl + 1).cuda()
def _fwd(xd_, ild_, cond_, tl1_):
    f, fl, _ = enc(xd_, ild_)
    return dec(f, cond_, fl, tl1_)[0]
del loss
gstep = _train.GraphedTrainStep(_fwd, list(enc.parameters()) + list(dec.parameters()),
                                dropout_streams=[enc.dropout_stream, dec.decoder.dropout_stream, dec.recognizer.dropout_stream])
loss_g, t_train_g = timed(lambda: gstep.step(xd, ild, cond, tl1))
print(f'transformer:32 N={N} joint-loss forward+backward (dropout 0.2), one HIP graph: {t_train_g*1e3:.2f} ms ({N/t_train_g:,.0f} utt/s), loss {loss_g.item():.4f}')
enc.eval(); dec.eval()


with torch.inference_mode():
    (feats, flen, _), t_enc = timed(lambda: enc(xd, ild))
    lp = dec.recognizer.log_probs(feats)
    (hyp_beam, _), t_beam = timed(lambda: beam.decode_batch(dec.recognizer.log_probs(feats), beam_size=16))
    (outs, olen, _, lps, _), t_dec = timed(lambda: dec.decode(feats, flen, tld))
print(f'transformer:32 N={N} math={math_mode}: encoder {t_enc*1e3:.2f} ms ({N/t_enc:,.0f} utt/s) | CTC beam16 {t_beam*1e3:.2f} ms '
      f'({N/t_beam:,.0f} utt/s) | greedy decode T={int(tl.max())+1} {t_dec*1e3:.2f} ms ({N/t_dec:,.0f} utt/s)')

# CPU oracle on the first NCPU utterances: same hypotheses?
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))   # the box share; os.cpu_count() reports the whole host
with torch.no_grad():
    t0 = time.perf_counter()
    f_ref, fl_ref = ref.audio_encoder_forward(pe, x[:NCPU], il[:NCPU], H)
    t_cpu_enc = time.perf_counter() - t0
    t0 = time.perf_counter()
    o_ref = ref.decoder_decode(pd, f_ref, fl_ref, tl[:NCPU], H, pre='decoder.')
    t_cpu_dec = time.perf_counter() - t0
    lp_ref = torch.nn.functional.linear(f_ref, pd['recognizer.classifier.weight'], pd['recognizer.classifier.bias']).log_softmax(-1)
    t0 = time.perf_counter()
    beam_ref = [lattice.ctc_beam_search_decode_logits(lp_ref[n], 16)[0][0] for n in range(NCPU)]
    t_cpu_beam = time.perf_counter() - t0
# decode of the sample alone on the GPU (greedy rows are independent except for the entropy monitor)
with torch.inference_mode():
    outs_s, olen_s, _, _, _ = dec.decode(feats[:NCPU], flen[:NCPU], tld[:NCPU])
hyp_gpu = [o.tolist() for o in outs_s.unbind()]
hyp_cpu = [o.tolist() for o in o_ref[0]]
errs = sum(edit_distance(a, b) for a, b in zip(hyp_gpu, hyp_cpu)); words = sum(len(b) for b in hyp_cpu)
berrs = sum(edit_distance(hyp_beam[n][0], beam_ref[n]) for n in range(NCPU)); bwords = sum(len(b) for b in beam_ref)
print(f'feature max-abs diff vs oracle {float((feats[:NCPU].cpu() - f_ref).abs().max()):.2e}; greedy WER vs CPU oracle {errs}/{words}; '
      f'beam-16 top hypothesis WER vs CPU oracle {berrs}/{bwords}')
print(f'CPU oracle ({torch.get_num_threads()} threads, {NCPU} utt): encoder {NCPU/t_cpu_enc:.1f} utt/s, greedy decode {NCPU/t_cpu_dec:.1f} utt/s, '
      f'beam16 {NCPU/t_cpu_beam:.1f} utt/s')

print(json.dumps({
    'metric': 'utterances/sec, encoder-decoder attention ASR `transformer:32` (BASELINE config 5): encoder, CTC beam-16, greedy attention decode',
    'value': round(N / (t_enc + t_dec), 1), 'unit': 'utterances/s', 'n_gpus': 1, 'dtype': {'bf16x3': 'bf16x3 (split-bf16 operands, 3 MFMAs per product, fp32 accumulate)', 'bf16': 'bf16', 'f32': 'f32'}[math_mode],
    'data': 'synthetic', 'config': {'workload': 'AudioEncoder 12L + CTCAttentionDecoder 12L, 8x64 heads, vocab 32, 80 frames x 80 mels',
                                    'batch': N, 'decode_steps': int(tl.max()) + 1, 'beam': 16, 'math': math_mode},
    'stages': {'encoder_utt_per_s': round(N / t_enc, 1), 'ctc_beam16_utt_per_s': round(N / t_beam, 1),
               'greedy_decode_utt_per_s': round(N / t_dec, 1), 'encoder_ms': round(t_enc * 1e3, 3), 'beam_ms': round(t_beam * 1e3, 3),
               'decode_ms': round(t_dec * 1e3, 3), 'train_fwd_bwd_ms': round(t_train * 1e3, 3), 'train_fwd_bwd_graph_ms': round(t_train_g * 1e3, 3),
               'train_fwd_bwd_utt_per_s': round(N / t_train, 1)},
    'wer_vs_cpu_oracle': {'greedy_errors': errs, 'greedy_words': words, 'beam_errors': berrs, 'beam_words': bwords,
                          'feature_max_abs_diff': float((feats[:NCPU].cpu() - f_ref).abs().max())},
    'cpu_baseline': {'value': round(NCPU / (t_cpu_enc + t_cpu_dec), 2), 'unit': 'utterances/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                     'sample': f'{NCPU} utterances: encoder {t_cpu_enc:.2f} s, greedy decode {t_cpu_dec:.2f} s, beam-16 {t_cpu_beam:.2f} s'},
}), flush=True)
