#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE itself on CPU.

Run in the build container only (the reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py

It imports /root/reference/ha (read-only) and records, for seeded inputs, what the reference
computes.  Only data (inputs, seeds, expected outputs) is written; no reference source.
Inputs/weights come from oracle.cpu_ref.make_params / synthetic_batch so the tests can
rebuild the large ones from a seed instead of storing 50 MB of weights.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')
sys.dont_write_bytecode = True

import types
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import ha.rnn, ha.recognizer, ha.ctc, ha.beam, ha.optim, ha.attention, ha.init, ha.transformer, ha.conv, ha.attention_audio   # the reference
import ha.star, ha.transducer

from oracle import cpu_ref, gpt_ref, transformer_ref, audio_encoder_ref

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def build_reference_model(enc_p, rec_p, F_, C, H, L, V):
    enc = ha.rnn.Encoder(input_dim=F_, subsample_dim=C, hidden_dim=H)
    if L != 3:
        enc.lstm = nn.LSTM(C, H, num_layers=L, batch_first=True, dropout=0.2)
    rec = ha.recognizer.TemporalClassifier(feat_dim=H, vocab_size=V)
    enc.load_state_dict(enc_p)
    rec.load_state_dict(rec_p)
    return enc, rec


def run_model_case(F_, C, H, L, V, B, T, S, seed, ragged):
    enc_p, rec_p = cpu_ref.make_params(F_, C, H, L, V, seed)
    x, il, tg, tl = cpu_ref.synthetic_batch(B, T, F_, V, S, seed)
    if ragged:
        il = torch.tensor([T - 3 * i for i in range(B)], dtype=torch.int64)
    enc, rec = build_reference_model(enc_p, rec_p, F_, C, H, L, V)
    enc.eval(); rec.eval()
    kept = {}

    def keep_logits(module, args, out):
        if out.requires_grad:
            out.retain_grad()
            kept['logits'] = out

    rec.classifier.register_forward_hook(keep_logits)
    feats, flen, _ = enc(x, il)
    feats.retain_grad()
    loss, _ = rec(feats, tg, flen, tl)
    loss.backward()
    with torch.no_grad():
        lp = rec.log_probs(feats)
        hyps, hlen, ali, scores, _ = rec.decode(feats, flen, tl)
    grads = {('encoder.' + k): p.grad for k, p in enc.named_parameters()}
    grads.update({('recognizer.' + k): p.grad for k, p in rec.named_parameters()})
    return dict(enc_p=enc_p, rec_p=rec_p, x=x, il=il, tg=tg, tl=tl, feats=feats.detach(), flen=flen,
                loss=loss.detach(), lp=lp, grads=grads, dlogits=kept['logits'].grad, dfeats=feats.grad,
                ali=ali, scores=scores, hyps=[[int(v) for v in h] for h in hyps], hlen=hlen)


def save_tiny():
    for name, L in (('g1_tiny_l2', 2), ('g1_tiny_l3', 3)):
        cfg = dict(F_=12, C=16, H=32, L=L, V=9, B=3, T=41, S=4, seed=7, ragged=True)
        r = run_model_case(**cfg)
        d = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
        for k, v in r['enc_p'].items():
            d['encoder.' + k] = v.numpy()
        for k, v in r['rec_p'].items():
            d['recognizer.' + k] = v.numpy()
        for k in ('x', 'il', 'tg', 'tl', 'feats', 'flen', 'loss', 'lp', 'dlogits', 'dfeats', 'ali', 'scores', 'hlen'):
            d[k] = r[k].numpy()
        for k, v in r['grads'].items():
            d['grad.' + k] = v.numpy()
        maxlen = max(1, max(len(h) for h in r['hyps']))
        d['hyps'] = np.array([h + [-1] * (maxlen - len(h)) for h in r['hyps']], dtype=np.int64)
        np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
        print(name, 'loss', float(r['loss']), 'hyps', r['hyps'])


def save_lc2x1024():
    cfg = dict(F_=80, C=128, H=1024, L=2, V=32, B=4, T=80, S=10, seed=42, ragged=True)
    r = run_model_case(**cfg)
    d = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
    for k in ('il', 'flen', 'loss', 'lp', 'dlogits', 'ali', 'scores', 'hlen'):
        d[k] = r[k].numpy()
    d['feats_slice'] = r['feats'][:, :, ::61].numpy()          # 17 of 1024 columns
    d['feats_sum'] = r['feats'].double().sum().numpy()
    d['dfeats_slice'] = r['dfeats'][:, :, ::61].numpy()
    for k, v in r['grads'].items():
        d['gradnorm.' + k] = v.double().norm().numpy()
        d['gradslice.' + k] = v.reshape(-1)[::9973].numpy()
    maxlen = max(1, max(len(h) for h in r['hyps']))
    d['hyps'] = np.array([h + [-1] * (maxlen - len(h)) for h in r['hyps']], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'g1_lc2x1024.npz'), **d)
    print('g1_lc2x1024 loss', float(r['loss']))


def save_lc2x1024_b64():
    """BASELINE config 2's grid: the same 2-layer H=1024 model at B=64 (256 step workgroups: 64 hidden tiles x 4 batch tiles),
    ragged input lengths (T' from 16 to 21).  Slices, sums and norms only: the full tensors would be 50 MB."""
    cfg = dict(F_=80, C=128, H=1024, L=2, V=32, B=64, T=80, S=10, seed=42)
    enc_p, rec_p = cpu_ref.make_params(cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'], cfg['seed'])
    x, il, tg, tl = cpu_ref.synthetic_batch(cfg['B'], cfg['T'], cfg['F_'], cfg['V'], cfg['S'], cfg['seed'])
    il = torch.tensor([cfg['T'] - 3 * (i % 8) for i in range(cfg['B'])], dtype=torch.int64)
    enc, rec = build_reference_model(enc_p, rec_p, cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'])
    enc.eval(); rec.eval()
    feats, flen, _ = enc(x, il)
    feats.retain_grad()
    loss, _ = rec(feats, tg, flen, tl)
    loss.backward()
    with torch.no_grad():
        lp = rec.log_probs(feats)
        hyps, hlen, ali, scores, _ = rec.decode(feats, flen, tl)
    d = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
    d.update(il=il.numpy(), flen=flen.numpy(), loss=loss.detach().numpy(), ali=ali.numpy(), hlen=hlen.numpy())
    d['feats_slice'] = feats.detach()[:, :, ::61].numpy()
    d['feats_sum'] = feats.detach().double().sum().numpy()
    d['dfeats_slice'] = feats.grad[:, :, ::61].numpy()
    d['lp_slice'] = lp[::3].numpy()
    grads = {('encoder.' + k): p.grad for k, p in enc.named_parameters()}
    grads.update({('recognizer.' + k): p.grad for k, p in rec.named_parameters()})
    for k, v in grads.items():
        d['gradnorm.' + k] = v.double().norm().numpy()
        d['gradslice.' + k] = v.reshape(-1)[::9973].numpy()
    maxlen = max(1, max(len(h) for h in hyps))
    d['hyps'] = np.array([[int(v) for v in h] + [-1] * (maxlen - len(h)) for h in hyps], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, 'g1_lc2x1024_b64.npz'), **d)
    print('g1_lc2x1024_b64 loss', float(loss))

    # three optimizer steps of the same model the way ha/loop.py:176-196 runs them (eval-mode dropout), on three seeded batches
    enc, rec = build_reference_model(enc_p, rec_p, cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'])
    enc.eval(); rec.eval()
    model = nn.ModuleDict({'encoder': enc, 'recognizer': rec})
    args = types.SimpleNamespace(lr=3e-4, weight_decay=0.01, beta1=0.9, beta2=0.99)
    opt = ha.optim.configure_optimizers(model, args, device_type='cpu', decay_lm_head=False)
    losses, gnorms = [], []
    t = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
    t['cfg_lr'] = np.array(args.lr)
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(cfg['B'], cfg['T'], cfg['F_'], cfg['V'], cfg['S'], 200 + step)
        feats, flen, _ = enc(x, il)
        loss, _ = rec(feats, tg, flen, tl)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 0.1, error_if_nonfinite=False)
        opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(float(loss)); gnorms.append(float(gn))
    t['losses'] = np.array(losses); t['gnorms'] = np.array(gnorms)
    for prefix, mod in (('encoder.', enc), ('recognizer.', rec)):
        for k, v in mod.state_dict().items():
            t['finalslice.' + prefix + k] = v.reshape(-1)[::4999].numpy()
            t['finalsum.' + prefix + k] = v.double().sum().numpy()
    np.savez_compressed(os.path.join(OUT, 'g1_train3_b64.npz'), **t)
    print('g1_train3_b64 losses', losses, 'gnorms', gnorms)


def save_stock3_b64():
    """The reference's STOCK encoder -- ha.rnn.Encoder(80, 128, 1024) with the 3-layer nn.LSTM it builds itself (ha/rnn.py:6-11; nothing
    swapped) + TemporalClassifier(1024, 32) -- at B = 64 with ragged lengths, eval mode: loss, feature / gradient slices, sums and norms
    (the secondary model; the weights are rebuilt from the seed, 86 MB would not fit a fixture)."""
    cfg = dict(F_=80, C=128, H=1024, L=3, V=32, B=64, T=80, S=10, seed=43)
    enc_p, rec_p = cpu_ref.make_params(cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'], cfg['seed'])
    x, il, tg, tl = cpu_ref.synthetic_batch(cfg['B'], cfg['T'], cfg['F_'], cfg['V'], cfg['S'], cfg['seed'])
    il = torch.tensor([cfg['T'] - 3 * (i % 8) for i in range(cfg['B'])], dtype=torch.int64)
    enc, rec = build_reference_model(enc_p, rec_p, cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'])
    assert enc.lstm.num_layers == 3                           # the reference's own stack, not a replacement
    enc.eval(); rec.eval()
    feats, flen, _ = enc(x, il)
    feats.retain_grad()
    loss, _ = rec(feats, tg, flen, tl)
    loss.backward()
    with torch.no_grad():
        lp = rec.log_probs(feats)
        hyps, hlen, ali, scores, _ = rec.decode(feats, flen, tl)
    d = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
    d.update(il=il.numpy(), flen=flen.numpy(), loss=loss.detach().numpy(), ali=ali.numpy(), hlen=hlen.numpy())
    d['feats_slice'] = feats.detach()[:, :, ::61].numpy()
    d['feats_sum'] = feats.detach().double().sum().numpy()
    d['dfeats_slice'] = feats.grad[:, :, ::61].numpy()
    d['lp_slice'] = lp[::3].numpy()
    grads = {('encoder.' + k): p.grad for k, p in enc.named_parameters()}
    grads.update({('recognizer.' + k): p.grad for k, p in rec.named_parameters()})
    for k, v in grads.items():
        d['gradnorm.' + k] = v.double().norm().numpy()
        d['gradslice.' + k] = v.reshape(-1)[::9973].numpy()
    np.savez_compressed(os.path.join(OUT, 'g1_stock3_b64.npz'), **d)
    print('g1_stock3_b64 loss', float(loss))


def save_train_steps():
    """Three optimizer steps the way ha/loop.py:176-196 runs them (eval-mode dropout)."""
    cfg = dict(F_=12, C=16, H=32, L=2, V=9, B=3, T=41, S=4, seed=11)
    enc_p, rec_p = cpu_ref.make_params(cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'], cfg['seed'])
    enc, rec = build_reference_model(enc_p, rec_p, cfg['F_'], cfg['C'], cfg['H'], cfg['L'], cfg['V'])
    enc.eval(); rec.eval()
    model = nn.ModuleDict({'encoder': enc, 'recognizer': rec})
    args = types.SimpleNamespace(lr=3e-3, weight_decay=0.01, beta1=0.9, beta2=0.99)
    opt = ha.optim.configure_optimizers(model, args, device_type='cpu', decay_lm_head=False)
    losses, gnorms = [], []
    d = {'cfg_' + k: np.array(v) for k, v in cfg.items()}
    d['cfg_lr'] = np.array(args.lr)
    for step in range(3):
        x, il, tg, tl = cpu_ref.synthetic_batch(cfg['B'], cfg['T'], cfg['F_'], cfg['V'], cfg['S'], 100 + step)
        feats, flen, _ = enc(x, il)
        loss, _ = rec(feats, tg, flen, tl)
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(enc.parameters(), 0.1, error_if_nonfinite=False)
        opt.step(); opt.zero_grad(set_to_none=True)
        losses.append(float(loss)); gnorms.append(float(gn))
    d['losses'] = np.array(losses); d['gnorms'] = np.array(gnorms)
    for k, v in enc.state_dict().items():
        d['final.encoder.' + k] = v.numpy()
    for k, v in rec.state_dict().items():
        d['final.recognizer.' + k] = v.numpy()
    np.savez_compressed(os.path.join(OUT, 'g1_train3.npz'), **d)
    print('g1_train3 losses', losses, 'gnorms', gnorms)


def ctc_case(name, T, N, C, S, seed, targets=None, il=None, tl=None):
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(T, N, C, generator=g).requires_grad_(True)
    em = logits.log_softmax(-1)
    if targets is None:
        targets = torch.randint(1, C, (N, S), generator=g)
    if il is None:
        il = torch.full((N,), T, dtype=torch.int64)
    if tl is None:
        tl = torch.full((N,), S, dtype=torch.int64)
    with torch.no_grad():
        nll3 = ha.ctc.ctc_forward_score3(em, targets, il, tl)
        red3 = ha.ctc.ctc_reduce_mean(nll3, tl)
    nll_t = F.ctc_loss(em, targets, il, tl, blank=0, reduction='none')
    mean_t = F.ctc_loss(em, targets, il, tl, blank=0, reduction='mean')
    finite = torch.isfinite(nll_t).all()
    if finite:
        mean_t.backward()
        dlogits = logits.grad.clone()
    else:
        dlogits = torch.zeros_like(logits)
    return {name + '.logits': logits.detach().numpy(), name + '.targets': targets.numpy(),
            name + '.il': il.numpy(), name + '.tl': tl.numpy(), name + '.score3': nll3.numpy(),
            name + '.reduce_mean3': red3.numpy(), name + '.torch_none': nll_t.detach().numpy(),
            name + '.torch_mean': mean_t.detach().numpy(), name + '.dlogits_mean': dlogits.numpy(),
            name + '.has_grad': np.array(bool(finite))}


def save_ctc():
    d = {}
    d.update(ctc_case('random', 21, 4, 32, 10, 0, tl=torch.tensor([10, 7, 5, 9])))
    d.update(ctc_case('repeat', 21, 3, 8, 6, 1, targets=torch.tensor([[3, 3, 3, 3, 3, 3], [1, 1, 2, 2, 1, 1], [5, 4, 4, 4, 5, 5]])))
    d.update(ctc_case('s1', 9, 2, 6, 1, 2))
    d.update(ctc_case('ragged', 21, 4, 32, 10, 3, il=torch.tensor([21, 17, 20, 12]), tl=torch.tensor([3, 10, 1, 6])))
    d.update(ctc_case('infeasible', 6, 2, 5, 4, 4, targets=torch.tensor([[1, 1, 1, 1], [1, 2, 3, 4]])))
    d.update(ctc_case('wide', 50, 2, 300, 40, 5, tl=torch.tensor([40, 33])))
    # the reference's own __main__ demo (ha/ctc.py:181-238): seed 2
    torch.manual_seed(2)
    l0 = torch.randn(5, 7).log_softmax(-1)
    t0 = torch.LongTensor([1, 2, 3, 3])
    d['demo.l0'] = l0.numpy(); d['demo.t0'] = t0.numpy()
    d['demo.score1'] = ha.ctc.ctc_forward_score1(l0, t0).numpy()
    d['demo.score2'] = ha.ctc.ctc_forward_score2(l0, t0).numpy()
    l1 = torch.randn(5, 7).log_softmax(-1)
    t1 = torch.LongTensor([1, 2, 3, 4])
    em = torch.stack([l0, l1], dim=1); tg = torch.stack([t0, t1], dim=0)
    d['demo.em'] = em.numpy(); d['demo.tg'] = tg.numpy()
    d['demo.score3'] = ha.ctc.ctc_forward_score3(em, tg, torch.LongTensor([5, 5]), torch.LongTensor([3, 4])).numpy()
    # long-T single-sequence cases where score1's wrap-around skip (ctc.py:29) shows
    g = torch.Generator().manual_seed(9)
    l2 = torch.randn(12, 5, generator=g).log_softmax(-1)
    t2 = torch.LongTensor([2, 4])
    d['wrap.l'] = l2.numpy(); d['wrap.t'] = t2.numpy()
    d['wrap.score1'] = ha.ctc.ctc_forward_score1(l2, t2).numpy()
    d['wrap.score2'] = ha.ctc.ctc_forward_score2(l2, t2).numpy()
    np.savez_compressed(os.path.join(OUT, 'g2_ctc.npz'), **d)
    print('g2_ctc demo', d['demo.score1'], d['demo.score2'], d['demo.score3'], 'wrap', d['wrap.score1'], d['wrap.score2'],
          'infeasible', d['infeasible.score3'], d['infeasible.torch_none'])


def pad_seqs(seqs):
    m = max(1, max(len(s) for s in seqs))
    return (np.array([list(s) + [-1] * (m - len(s)) for s in seqs], dtype=np.int64),
            np.array([len(s) for s in seqs], dtype=np.int64))


def save_beam():
    d = {}
    probs = F.one_hot(torch.tensor([0, 3, 1, 2, 2, 0, 0, 2, 0, 0, 0, 1, 2, 3])).float()   # ha/beam.py:144
    seqs, sc = ha.beam.ctc_beam_search_decode_logits(torch.log(probs))
    d['onehot.logits'] = torch.log(probs).numpy()
    d['onehot.seqs'], d['onehot.lens'] = pad_seqs(seqs); d['onehot.scores'] = sc.numpy(); d['onehot.beam'] = np.array(3)
    for name, T, V, beam, seed in (('r21x32b16', 21, 32, 16, 0), ('r21x32b3', 21, 32, 3, 0), ('r6x4b4', 6, 4, 4, 0),
                                   ('r30x9b5', 30, 9, 5, 3), ('r21x32b33', 21, 32, 33, 5)):
        g = torch.Generator().manual_seed(seed)
        lg = torch.randn(T, V, generator=g).log_softmax(-1)
        seqs, sc = ha.beam.ctc_beam_search_decode_logits(lg, beam_size=beam)
        d[name + '.logits'] = lg.numpy(); d[name + '.beam'] = np.array(beam)
        d[name + '.seqs'], d[name + '.lens'] = pad_seqs(seqs); d[name + '.scores'] = sc.numpy()
    # peaked emissions (what a trained model produces): rankings never hinge on sub-ulp ties
    for name, T, V, beam, seed, scale in (('p21x32b16', 21, 32, 16, 10, 6.0), ('p40x32b8', 40, 32, 8, 11, 8.0),
                                          ('p21x256b4', 21, 256, 4, 12, 6.0), ('p64x9b9', 64, 9, 9, 13, 5.0)):
        g = torch.Generator().manual_seed(seed)
        lg = (torch.randn(T, V, generator=g) * scale).log_softmax(-1)
        seqs, sc = ha.beam.ctc_beam_search_decode_logits(lg, beam_size=beam)
        d[name + '.logits'] = lg.numpy(); d[name + '.beam'] = np.array(beam)
        d[name + '.seqs'], d[name + '.lens'] = pad_seqs(seqs); d[name + '.scores'] = sc.numpy()
    # probability-domain twin: NameError as shipped (beam.py:46) ...
    try:
        ha.beam.ctc_beam_search_decode_probs(probs)
        d['probs.raises_nameerror'] = np.array(False)
    except NameError:
        d['probs.raises_nameerror'] = np.array(True)
    # ... and what it computes once the missing module global exists
    ha.beam.device = 'cpu'
    g = torch.Generator().manual_seed(1)
    pr = torch.randn(12, 6, generator=g).softmax(-1)
    seqs, sc = ha.beam.ctc_beam_search_decode_probs(pr, beam_size=4)
    d['probs.probs'] = pr.numpy(); d['probs.beam'] = np.array(4)
    d['probs.seqs'], d['probs.lens'] = pad_seqs(seqs); d['probs.scores'] = sc.numpy()
    seqs, sc = ha.beam.ctc_beam_search_decode_probs(probs)
    d['probs_onehot.seqs'], d['probs_onehot.lens'] = pad_seqs(seqs); d['probs_onehot.scores'] = sc.numpy()
    del ha.beam.device
    # beam wider than 1+V at t=0 -> topk raises RuntimeError
    try:
        ha.beam.ctc_beam_search_decode_logits(torch.zeros(3, 4).log_softmax(-1), beam_size=6)
        d['toowide.raises'] = np.array(False)
    except RuntimeError:
        d['toowide.raises'] = np.array(True)
    np.savez_compressed(os.path.join(OUT, 'g3_beam.npz'), **d)
    print('g3_beam onehot', d['onehot.seqs'].tolist(), d['onehot.scores'], 'r6x4b4', d['r6x4b4.seqs'][0], d['r6x4b4.scores'][0])


def gpt_case(name, vocab, block, n_layer, n_head, n_embd, bias, B, T, seed, store_params, stable=False, causal=True):
    cfg = ha.init.GPTConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head, n_embd=n_embd, bias=bias,
                            stable_embedding=stable, causal=causal)
    model = ha.attention.GPT(cfg).eval()
    params = gpt_ref.make_gpt_params(vocab, block, n_layer, n_head, n_embd, bias, seed, stable=stable)
    missing = model.load_state_dict(params, strict=True)
    inputs, targets = gpt_ref.synthetic_tokens(B, T, vocab, seed + 1)
    with torch.no_grad():
        per_tok = model.forward_all(inputs, targets, reduction='none')
        mean = model.forward_all(inputs, targets, reduction='mean')
    d = {'cfg': np.array([vocab, block, n_layer, n_head, n_embd, int(bias), B, T, seed]), 'stable': np.array(int(stable)),
         'causal': np.array(int(causal)),
         'inputs': inputs.numpy(), 'targets': targets.numpy(), 'per_token': per_tok.numpy(), 'mean': mean.numpy()}
    # the training direction (ha/attention_loop.py:203-208): gradients of the mean loss w.r.t. every parameter
    model.zero_grad()
    model.forward_all(inputs, targets, reduction='mean').backward()
    for k, v in model.named_parameters():
        if store_params:
            d['grad.' + k] = v.grad.numpy()
        else:                                     # large model: norms of every gradient and a strided sample of each
            d['gradnorm.' + k] = np.array(float(v.grad.norm()))
            d['gradsample.' + k] = v.grad.flatten()[::max(1, v.numel() // 1000)][:1000].numpy().copy()
    if store_params:
        for k, v in params.items():
            d['param.' + k] = v.numpy()
        # generation path (ha/attention.py:253-279): prefill, then two single-token continuations through the KV cache
        with torch.no_grad() if causal else torch.no_grad():
            split = T // 2
            logits0, past = model(inputs[:, :split])
            logits1, past1 = model(inputs[:, split:split + 1], past=past)
            logits2, past2 = model(inputs[:, split + 1:split + 4], past=past1)
        d['gen.split'] = np.array(split)
        d['gen.logits0'], d['gen.logits1'], d['gen.logits2'] = logits0.numpy(), logits1.numpy(), logits2.numpy()
        d['gen.present2'] = past2.numpy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'mean nats/token', float(mean))


def audio_encoder_case(name, d_input, n_embd, n_head, n_layer, block, bias, vocab, B, T, S, seed, full_grads):
    """ha.attention_audio.AudioEncoder without rotary embeddings (the `audio-encoder` arch, ha/init.py:132-139) + the CTC head it is
    trained with: features, lengths, loss, and the gradient of every parameter."""
    cfg = ha.init.AudioEncoderConfig(block_size=block, vocab_size=vocab, n_layer=n_layer, n_head=n_head, n_embd=n_embd, bias=bias,
                                     d_input=d_input)
    cfg.rotary_emb_dim = 0
    enc = ha.attention_audio.AudioEncoder(cfg).eval()
    params = audio_encoder_ref.make_params(d_input, n_embd, n_layer, block, bias, seed)
    enc.load_state_dict(params, strict=True)
    assert not enc.transformer.wpe.weight.requires_grad
    rec = ha.recognizer.TemporalClassifier(feat_dim=n_embd, vocab_size=vocab).eval()
    rec_p, x, il, tg, tl = audio_encoder_ref.make_head_and_batch(n_embd, vocab, d_input, B, T, S, seed)
    rec.load_state_dict(rec_p)
    feats, flen, stats = enc(x, il)
    feats.retain_grad()
    loss, _ = rec(feats, tg, flen, tl)
    loss.backward()
    d = {'cfg': np.array([d_input, n_embd, n_head, n_layer, block, int(bias), vocab, B, T, S, seed]),
         'feats': feats.detach().numpy(), 'flen': flen.numpy(), 'loss': loss.detach().numpy(),
         'dfeats': feats.grad.numpy(), 'wpe_head': enc.transformer.wpe.weight[:8].detach().numpy()}
    grads = {('grad.' + k): v.grad for k, v in enc.named_parameters() if v.grad is not None}
    grads.update({('recgrad.' + k): v.grad for k, v in rec.named_parameters()})
    for k, v in grads.items():
        if full_grads:
            d[k] = v.numpy()
        else:
            d['norm.' + k] = np.array(float(v.double().norm()))
            d['slice.' + k] = v.reshape(-1)[::97].numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'loss', float(loss), 'flen', flen.tolist())


def save_audio_encoder():
    audio_encoder_case('g7_audio_encoder_tiny', 20, 64, 2, 2, 64, False, 11, 3, 41, 4, 21, True)
    audio_encoder_case('g7_audio_encoder_bias', 80, 128, 4, 3, 128, True, 32, 4, 80, 10, 22, False)


def save_gpt():
    gpt_case('g5_gpt_tiny_nobias', 97, 32, 2, 2, 64, False, 3, 20, 5, True)
    gpt_case('g5_gpt_tiny_bias', 97, 48, 3, 1, 64, True, 2, 40, 6, True)
    gpt_case('g5_gpt_tiny_stable', 61, 32, 2, 2, 64, False, 3, 24, 8, True, stable=True)
    gpt_case('g5_gpt_tiny_bidir', 71, 40, 2, 2, 64, True, 2, 33, 9, True, causal=False)     # MLM / audio-encoder blocks
    gpt_case('g5_gpt2_small', 50304, 1024, 12, 12, 768, False, 1, 1024, 7, False)   # params rebuilt from the seed

def asr_case(name, vocab, head_dim, heads, enc_layers, dec_layers, conv_dim, N, T, S, seed, strides=(2, 2, 2), F_=80):
    """ha.transformer AudioEncoder + CTCAttentionDecoder (the `transformer:V` family of ha/init.py:234-239) on CPU:
    encoder features/lengths, teacher-forced losses, joint CTC loss, and the greedy decode under fp16 autocast
    (the only way the reference's decode runs: its KV caches are float16)."""
    enc = ha.transformer.AudioEncoder(head_dim=head_dim, heads=heads, layers=enc_layers, p_drop=0.2, input_dim=F_,
                                      conv_dim=conv_dim, conv_strides=strides).eval()
    dec = ha.transformer.CTCAttentionDecoder(vocab=vocab, head_dim=head_dim, heads=heads, p_drop=0.2, layers=dec_layers).eval()
    pe = transformer_ref.make_encoder_params(head_dim, heads, enc_layers, F_, conv_dim, len(strides), seed)
    pd = transformer_ref.make_decoder_params(vocab, head_dim, heads, dec_layers, seed + 1)
    enc.load_state_dict(pe, strict=True)
    dec.load_state_dict(pd, strict=True)
    x, il, tg, tl = transformer_ref.synthetic_asr_batch(N, T, F_, vocab, S, seed + 2)
    d = {'cfg': np.array([vocab, head_dim, heads, enc_layers, dec_layers, conv_dim, N, T, S, seed, F_]), 'strides': np.array(strides)}
    with torch.no_grad():
        feats, flen, _ = enc(x, il)
        d['features'], d['feature_lengths'] = feats.numpy(), flen.numpy()
        for red in ('mean', 'none', 'sumeach'):
            loss, _ = dec.decoder(feats, tg, flen, tl, reduction=red, drop_labels=False)
            d['decoder_loss.' + red] = loss.numpy()
        cond = torch.cat([torch.full((N, 1), 5, dtype=torch.long), tg], dim=1)       # one prompt token in front
        joint, _ = dec(feats, cond, flen, tl + 1)
        d['joint_loss'] = joint.numpy()
        _, stats = dec.decoder(feats, tg, flen, tl, measure_entropy=True, drop_labels=False)
        d['meme_entropy'] = torch.stack(stats['meme_entropy']).numpy()
        d['self_entropy'] = torch.stack(stats['self_entropy']).numpy()
        with torch.autocast('cpu', dtype=torch.float16):
            outs, olen, _, lps, ents = dec.decode(feats, flen, tl)
        d['decode.tokens'], d['decode.token_lens'] = pad_seqs([o.tolist() for o in outs.unbind()])
        d['decode.output_lengths'], d['decode.log_probs'], d['decode.sum_entropies'] = olen.numpy(), lps.numpy(), ents.numpy()
        user = torch.tensor([[7, 9]] * N)
        with torch.autocast('cpu', dtype=torch.float16):
            outs, olen, _, lps, ents = dec.decode(feats, flen, tl, prompt=user)
        d['decode_prompt.tokens'], d['decode_prompt.token_lens'] = pad_seqs([o.tolist() for o in outs.unbind()])
        d['decode_prompt.output_lengths'], d['decode_prompt.log_probs'] = olen.numpy(), lps.numpy()
        # decisiveness of the fixture: smallest top-2 margin of the oracle's own (fp32, fp16-cache) greedy run
        o = transformer_ref.decoder_decode(pd, feats, flen, tl, heads, pre='decoder.')
        top2 = o[4].topk(2, dim=-1).values
        d['decode.min_margin'] = (top2[..., 0] - top2[..., 1]).min().numpy()
    # the training direction (eval mode, so no dropout / label dropout): encoder -> joint loss -> backward, every parameter
    enc.zero_grad(); dec.zero_grad()
    feats, flen, _ = enc(x, il)
    joint, _ = dec(feats, cond, flen, tl + 1)
    joint.backward()
    d['train.joint_loss'] = joint.detach().numpy()
    small = sum(p.numel() for p in list(enc.parameters()) + list(dec.parameters())) < 400000
    for pre, m in (('encoder.', enc), ('decoder.', dec)):
        for k, v in m.named_parameters():
            if small:
                d['grad.' + pre + k] = v.grad.numpy()
            else:
                d['gradnorm.' + pre + k] = np.array(float(v.grad.norm()))
                d['gradsample.' + pre + k] = v.grad.flatten()[::max(1, v.numel() // 500)][:500].numpy().copy()
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **d)
    print(name, 'flen', flen.tolist(), 'dec loss', float(d['decoder_loss.mean']), 'joint', float(joint), 'decode lens',
          d['decode.output_lengths'].tolist(), 'margin', float(d['decode.min_margin']))


def save_asr():
    # seeds picked (scan over the oracle's greedy run) so that greedy choices are decisive (top-2 margin > 0.15 on every
    # alive step: fp16-vs-fp32 rounding cannot flip a token), rows differ, and some rows stop early on ETX
    asr_case('g6_asr_tiny', 32, 16, 2, 2, 2, 32, 3, 80, 6, 125)
    asr_case('g6_asr_tiny_s221', 40, 32, 1, 1, 2, 48, 4, 53, 5, 89, strides=(2, 2, 1), F_=40)
    asr_case('g6_asr_tiny_stop', 40, 32, 1, 1, 2, 48, 4, 53, 5, 65, strides=(2, 2, 1), F_=40)
    asr_case('g6_asr_transformer32', 32, 64, 8, 12, 12, 256, 2, 80, 8, 28)          # ha/init.py:234-239 `transformer:32`
    asr_case('g6_asr_transformer32_stop', 32, 64, 8, 12, 12, 256, 2, 80, 8, 44)
    d = {}
    g = torch.Generator().manual_seed(3)
    for nm, shape, t0 in (('a', (2, 3, 7, 16), 0), ('b', (1, 2, 1, 64), 11), ('c', (5, 8), 3)):
        x = torch.randn(shape, generator=g)
        d[f'rope.{nm}.x'], d[f'rope.{nm}.t0'], d[f'rope.{nm}.y'] = x.numpy(), np.array(t0), ha.transformer.rotate_interleaved(x, t0=t0).numpy()
    q, k, v = (torch.randn(2, 3, 9, 16, generator=g) for _ in range(3))
    mask = torch.rand(2, 1, 1, 9, generator=g) > 0.7
    mask[..., 0] = False
    y, ent = ha.transformer.attend(q, k, v, mask)
    d['attend.q'], d['attend.k'], d['attend.v'], d['attend.mask'], d['attend.y'], d['attend.entropy'] = (
        q.numpy(), k.numpy(), v.numpy(), mask.numpy(), y.numpy(), ent.numpy())
    conv = ha.conv.ConvEncoder(input_dim=8, hidden_dim=8, output_dim=8, strides=(2, 2, 2))
    lens = torch.arange(1, 200)
    d['lengths.in'], d['lengths.s222'] = lens.numpy(), conv.subsampled_lengths(lens).numpy()
    conv = ha.conv.ConvEncoder(input_dim=8, hidden_dim=8, output_dim=8, strides=(2, 2, 1))
    d['lengths.s221'] = conv.subsampled_lengths(lens).numpy()
    np.savez_compressed(os.path.join(OUT, 'g6_asr_parts.npz'), **d)


def save_rnn_decoder():
    """ha.rnn.Decoder (the `hal` LSTM language model, ha/rnn.py:30-77) as the reference runs it: two TBPTT chunks with the carried,
    detached state (rnnlm.py:191-211 pattern), cross-entropy on the tied output layer, every gradient of the second chunk."""
    V, E, L, T, N = 50, 64, 2, 9, 3
    torch.manual_seed(7)
    dec = ha.rnn.Decoder(V, E, E, L).eval()
    g = torch.Generator().manual_seed(8)
    tokens = torch.randint(0, V, (2, T + 1, N), generator=g)
    d = {'cfg': np.array([V, E, L, T, N]), 'tokens': tokens.numpy()}
    for k, v in dec.state_dict().items():
        d['param.' + k] = v.numpy()
    state = dec.init_hidden(N)
    for chunk in range(2):
        dec.zero_grad()
        logits, state = dec(tokens[chunk, :-1], dec.truncate_hidden(state))
        loss = F.cross_entropy(logits, tokens[chunk, 1:].reshape(-1))
        loss.backward()
        d[f'chunk{chunk}.logits'] = logits.detach().numpy()
        d[f'chunk{chunk}.h'], d[f'chunk{chunk}.c'] = state[0].detach().numpy(), state[1].detach().numpy()
        d[f'chunk{chunk}.loss'] = loss.detach().numpy()
    for k, p_ in dec.named_parameters():
        d['grad.' + k] = p_.grad.numpy()
    lbf, _ = dec.forward_batch_first(tokens[0, :-1].t(), dec.init_hidden(N))
    d['batch_first.logits'] = lbf.detach().numpy()
    np.savez_compressed(os.path.join(OUT, 'g10_rnn_decoder.npz'), **d)
    print('g10_rnn_decoder losses', d['chunk0.loss'], d['chunk1.loss'])


def star_case(name, T, N, C, S, seed, penalty, targets=None, il=None, tl=None):
    """ha.star.star_ctc_forward_score on seeded log-probabilities + its autograd gradient w.r.t. the emissions (sum of losses)."""
    g = torch.Generator().manual_seed(seed)
    em = torch.randn(T, N, C, generator=g).log_softmax(-1).requires_grad_(True)
    if targets is None:
        targets = torch.randint(1, C, (N, S), generator=g)
    il = torch.full((N,), T, dtype=torch.int64) if il is None else il
    tl = torch.full((N,), S, dtype=torch.int64) if tl is None else tl
    losses = ha.star.star_ctc_forward_score(em, targets, il, tl, star_penalty=penalty)
    losses.sum().backward()
    with torch.no_grad():
        s_em, s_tg = ha.star.intersperse_stars(em, targets)
    return {name + '.emissions': em.detach().numpy(), name + '.targets': targets.numpy(), name + '.il': il.numpy(), name + '.tl': tl.numpy(),
            name + '.penalty': np.float32(penalty), name + '.losses': losses.detach().numpy(), name + '.grad': em.grad.numpy(),
            name + '.star_targets': s_tg.numpy(), name + '.star_emissions_t0': s_em[0].numpy()}


def save_star():
    d = {}
    d.update(star_case('random', 21, 4, 32, 10, 0, -0.5, il=torch.tensor([21, 17, 20, 12]), tl=torch.tensor([10, 7, 5, 9])))
    d.update(star_case('repeat', 21, 3, 8, 6, 1, -0.5, targets=torch.tensor([[3, 3, 3, 3, 3, 3], [1, 1, 2, 2, 1, 1], [5, 4, 4, 4, 5, 5]])))
    d.update(star_case('s1', 9, 2, 6, 1, 2, -2.0))
    d.update(star_case('nopenalty', 12, 3, 10, 4, 3, 0.0, tl=torch.tensor([4, 2, 3])))
    d.update(star_case('padded', 15, 3, 9, 5, 4, -0.5, targets=torch.tensor([[2, 5, 0, 0, 0], [1, 1, 7, 0, 0], [8, 3, 3, 2, 1]]),
                       il=torch.tensor([15, 9, 15]), tl=torch.tensor([2, 3, 5])))
    # the reference's own __main__ demo (ha/star.py:199-215): seed 2, penalty -100
    torch.manual_seed(2)
    logits = torch.stack([torch.randn(10, 7).log_softmax(-1), torch.randn(10, 7).log_softmax(-1)], dim=1)
    targets = torch.tensor([[1, 2, 3, 3], [1, 2, 3, 4]])
    il, tl = torch.LongTensor([5, 10]), torch.LongTensor([3, 4])
    d['demo.emissions'], d['demo.targets'], d['demo.il'], d['demo.tl'] = logits.numpy(), targets.numpy(), il.numpy(), tl.numpy()
    d['demo.penalty'] = np.float32(-100)
    d['demo.losses'] = ha.star.star_ctc_forward_score(logits, targets, il, tl, star_penalty=-100).numpy()
    np.savez_compressed(os.path.join(OUT, 'g8_star.npz'), **d)
    print('g8_star demo', d['demo.losses'], 'random', d['random.losses'], 'padded', d['padded.losses'])


def transducer_case(name, N, T, U, K, seed, jl=None, tl=None):
    """ha.transducer.transducer_forward_score on a seeded joint + its autograd gradient w.r.t. the joint (sum of losses)."""
    g = torch.Generator().manual_seed(seed)
    f, p = torch.randn(N, T, K, generator=g), torch.randn(N, U + 1, K, generator=g)
    joint = (f[:, :, None, :] + p[:, None, :, :]).log_softmax(dim=-1).requires_grad_(True)
    targets = torch.randint(0, K, (N, U), generator=g)                  # label 0 (= the blank id) does occur, as in the reference's tests
    jl = torch.full((N,), T, dtype=torch.int32) if jl is None else jl
    tl = torch.full((N,), U, dtype=torch.int32) if tl is None else tl
    losses = ha.transducer.transducer_forward_score(joint, targets, jl, tl)
    losses.sum().backward()
    return {name + '.joint': joint.detach().numpy(), name + '.targets': targets.numpy(), name + '.jl': jl.numpy(), name + '.tl': tl.numpy(),
            name + '.losses': losses.detach().numpy(), name + '.grad': joint.grad.numpy()}


def save_transducer():
    d = {}
    d.update(transducer_case('batched', 13, 7, 4, 6, 42))                   # the shape of ha/transducer.py:210-231 (test_batched)
    # T must lie in (2^(k-1/2), 2^k]: the reference pads its scan to 2 ** round(log2(T)) (ha/transducer.py:194) and raises otherwise
    d.update(transducer_case('ragged', 5, 24, 10, 32, 1, jl=torch.tensor([24, 17, 20, 12, 1], dtype=torch.int32),
                             tl=torch.tensor([10, 7, 0, 9, 3], dtype=torch.int32)))
    d.update(transducer_case('long', 2, 64, 33, 8, 2, jl=torch.tensor([64, 50], dtype=torch.int32), tl=torch.tensor([33, 20], dtype=torch.int32)))
    # single-sequence flood-fill variant (ha/transducer.py:145-172) on sequence 0 of 'batched'
    j0, t0 = torch.from_numpy(d['batched.joint'][0]), torch.from_numpy(d['batched.targets'][0])
    d['batched.score4_seq0'] = ha.transducer.transducer_forward_score4(j0, t0).numpy()
    np.savez_compressed(os.path.join(OUT, 'g9_transducer.npz'), **d)
    print('g9_transducer batched', d['batched.losses'][:4], 'score4', d['batched.score4_seq0'], 'ragged', d['ragged.losses'])


def save_fbank():
    """g11_fbank: the 80-mel kaldi filterbank of ha/data.py:136-140 (torchaudio.compliance.kaldi.fbank(wav, num_mel_bins=80)).
    torchaudio is not installed here, so the vectors come from the Hugging Face transformers port of that function -- the numpy
    path Speech2TextFeatureExtractor._extract_fbank_features takes when torchaudio is absent (transformers.audio_utils.spectrogram
    with the povey window, kaldi mel scale, pre-emphasis 0.97, DC removal, log floored at float32 eps), which transformers keeps
    equal to ta_kaldi.fbank.  Inputs: waveforms in the 16-bit integer range torchaudio's callers feed (a tone + noise + offset, white
    noise, a chirp, a 401-sample clip = one frame); the port multiplies by 2**15 itself, so it gets wav / 2**15."""
    import transformers
    from transformers import Speech2TextFeatureExtractor
    fe = Speech2TextFeatureExtractor(feature_size=80, num_mel_bins=80, sampling_rate=16000, dither=0.0)
    rng = np.random.default_rng(11)
    t = np.arange(16000) / 16000.0
    waves = {
        'tone': 0.3 * np.sin(2 * np.pi * 440.0 * t) + 0.05 * rng.standard_normal(16000) + 0.01,
        'noise': 0.1 * rng.standard_normal(12345),
        'chirp': 0.5 * np.sin(2 * np.pi * (100.0 + 3000.0 * t[:8000]) * t[:8000]),
        'one_frame': 0.2 * rng.standard_normal(401),
    }
    d = {'transformers_version': np.array(transformers.__version__)}
    for k, w in waves.items():
        w16 = (w * 2 ** 15).astype(np.float32)                          # what torchaudio.compliance.kaldi.fbank is handed
        d[f'{k}.wav'] = w16
        d[f'{k}.fbank'] = fe._extract_fbank_features((w16 / 2 ** 15).astype(np.float32)).astype(np.float32)
        print('g11_fbank', k, d[f'{k}.fbank'].shape, float(np.abs(d[f'{k}.fbank']).max()))
    np.savez_compressed(os.path.join(OUT, 'g11_fbank.npz'), **d)


if __name__ == '__main__':
    if len(sys.argv) > 1:                       # e.g. `make_golden.py save_lc2x1024_b64`: regenerate the named fixtures only
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    save_fbank()
    save_star()
    save_transducer()
    save_rnn_decoder()
    save_lc2x1024_b64()
    save_stock3_b64()
    save_audio_encoder()
    save_asr()
    save_gpt()
    save_tiny()
    save_train_steps()
    save_ctc()
    save_beam()
    save_lc2x1024()
