"""CPU restatement of haloop's LSTM-CTC acoustic path on stock torch CPU ops.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's arithmetic for this path *is* third-party torch (nn.Conv1d, nn.LSTM,
nn.Linear, log_softmax, F.ctc_loss, clip_grad_norm_, AdamW -- SURVEY.md section 8c), so the
restatement composes the same ATen CPU operators, laid out functionally (explicit parameter
dicts keyed by the reference's state-dict names, explicit optional dropout masks) instead of
as the reference's module classes.  Pinned against the imported reference by
tests/golden/*.npz (generator: tests/golden/make_golden.py).

State-dict names (ha/rnn.py:9,11; ha/recognizer.py:40):
    encoder:    subsample.weight [C,F,5]  subsample.bias [C]
                lstm.weight_ih_l{k} [4H,in]  lstm.weight_hh_l{k} [4H,H]  lstm.bias_ih_l{k}  lstm.bias_hh_l{k}
    recognizer: classifier.weight [V,H]  classifier.bias [V]
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from . import philox

CONV_KERNEL, CONV_STRIDE, CONV_PAD = 5, 4, 3            # ha/rnn.py:9

# dropout stream ids shared with haloop_amd (csrc/philox.h)
STREAM_SUBSAMPLE = 1
STREAM_LSTM_LAYER0 = 16          # + layer index
STREAM_CLASSIFIER = 2


def subsampled_lengths(input_lengths):
    """floor((len + 2p - k) / s + 1) as int32, through float division like ha/rnn.py:13-18."""
    o = input_lengths + 2 * CONV_PAD - CONV_KERNEL
    return torch.floor(o / CONV_STRIDE + 1).int()


def make_params(input_dim, subsample_dim, hidden_dim, num_layers, vocab_size, seed):
    """Deterministic parameter set (own generator; independent of torch's module init order).

    Uniform(-1/sqrt(fan), 1/sqrt(fan)) like torch's defaults, drawn per tensor from a seeded
    CPU generator so that the same weights can be rebuilt on any box from ``seed`` alone.
    """
    g = torch.Generator().manual_seed(seed)

    def u(shape, fan):
        bound = 1.0 / math.sqrt(fan)
        return (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound

    enc = OrderedDict()
    enc['subsample.weight'] = u((subsample_dim, input_dim, CONV_KERNEL), input_dim * CONV_KERNEL)
    enc['subsample.bias'] = u((subsample_dim,), input_dim * CONV_KERNEL)
    for k in range(num_layers):
        in_dim = subsample_dim if k == 0 else hidden_dim
        enc[f'lstm.weight_ih_l{k}'] = u((4 * hidden_dim, in_dim), hidden_dim)
        enc[f'lstm.weight_hh_l{k}'] = u((4 * hidden_dim, hidden_dim), hidden_dim)
        enc[f'lstm.bias_ih_l{k}'] = u((4 * hidden_dim,), hidden_dim)
        enc[f'lstm.bias_hh_l{k}'] = u((4 * hidden_dim,), hidden_dim)
    rec = OrderedDict()
    rec['classifier.weight'] = u((vocab_size, hidden_dim), hidden_dim)
    rec['classifier.bias'] = u((vocab_size,), hidden_dim)
    return enc, rec


def num_lstm_layers(enc):
    return sum(1 for k in enc if k.startswith('lstm.weight_hh_l'))


def philox_masks(B, Tp, subsample_dim, hidden_dim, num_layers, p_enc, p_cls, seed, offset):
    """The masks the HIP path applies in training mode (element order = its buffer layouts).

    subsample output and inter-layer LSTM outputs are indexed time-major [T', B, dim]
    (haloop_amd keeps them time-major); the classifier input is indexed [B, T', H].
    Returned batch-first for use below.
    """
    m = {}
    x = philox.dropout_mask(Tp * B * subsample_dim, p_enc, seed, STREAM_SUBSAMPLE, offset)
    m['subsample'] = torch.from_numpy(x).view(Tp, B, subsample_dim).transpose(0, 1).contiguous()
    for k in range(num_layers - 1):
        x = philox.dropout_mask(Tp * B * hidden_dim, p_enc, seed, STREAM_LSTM_LAYER0 + k, offset)
        m[f'lstm{k}'] = torch.from_numpy(x).view(Tp, B, hidden_dim).transpose(0, 1).contiguous()
    x = philox.dropout_mask(B * Tp * hidden_dim, p_cls, seed, STREAM_CLASSIFIER, offset)
    m['classifier'] = torch.from_numpy(x).view(B, Tp, hidden_dim)
    return m


def _lstm_layer(x, w_ih, w_hh, b_ih, b_hh, state=None, batch_first=True):
    """One nn.LSTM layer through ATen's lstm (same kernel nn.LSTM dispatches to)."""
    B = x.shape[0] if batch_first else x.shape[1]
    H = w_hh.shape[1]
    if state is None:
        h0 = x.new_zeros(1, B, H)
        c0 = x.new_zeros(1, B, H)
    else:
        h0, c0 = state
    out, hn, cn = torch._VF.lstm(x, (h0, c0), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False, batch_first)
    return out, (hn, cn)


def encoder_forward(enc, inputs, input_lengths, masks=None, p_torch=0.0):
    """relu(LSTM(drop(relu(conv(x^T)^T))))  --  ha/rnn.py:20-26.

    masks: None (eval), or dict from philox_masks (explicit multiplicative masks).
    p_torch > 0 uses torch's own RNG like the reference does in train() mode.
    """
    x = F.conv1d(inputs.mT, enc['subsample.weight'], enc['subsample.bias'],
                 stride=CONV_STRIDE, padding=CONV_PAD).mT
    x = x.relu()
    if masks is not None:
        x = x * masks['subsample']
    elif p_torch > 0:
        x = F.dropout(x, p_torch, True)
    L = num_lstm_layers(enc)
    for k in range(L):
        x, _ = _lstm_layer(x, enc[f'lstm.weight_ih_l{k}'], enc[f'lstm.weight_hh_l{k}'],
                           enc[f'lstm.bias_ih_l{k}'], enc[f'lstm.bias_hh_l{k}'])
        if k < L - 1:
            if masks is not None:
                x = x * masks[f'lstm{k}']
            elif p_torch > 0:
                x = F.dropout(x, p_torch, True)
    return x.relu(), subsampled_lengths(input_lengths), {}


def classifier_log_probs(rec, features, mask=None, p_torch=0.0):
    """log_softmax(Linear(dropout(f)))  --  ha/recognizer.py:43-46."""
    if mask is not None:
        features = features * mask
    elif p_torch > 0:
        features = F.dropout(features, p_torch, True)
    return F.linear(features, rec['classifier.weight'], rec['classifier.bias']).log_softmax(dim=-1)


def classifier_loss(rec, features, targets, input_lengths=None, target_lengths=None, mask=None, p_torch=0.0):
    """F.ctc_loss(reduction='mean', blank=0) on time-major log-probs -- ha/recognizer.py:61-73."""
    if input_lengths is None:
        input_lengths = torch.full((features.shape[0],), features.shape[1], dtype=torch.long)
    if target_lengths is None:
        target_lengths = torch.full((features.shape[0],), len(targets), dtype=torch.long)
    lp = classifier_log_probs(rec, features, mask, p_torch).to(torch.float32).permute(1, 0, 2)
    return F.ctc_loss(lp, targets, input_lengths=input_lengths, target_lengths=target_lengths), {}


def lstm_ctc_loss(enc, rec, x, input_lengths, targets, target_lengths, masks=None, p_torch=0.0):
    feats, flen, _ = encoder_forward(enc, x, input_lengths, masks, p_torch)
    cmask = None if masks is None else masks['classifier']
    loss, _ = classifier_loss(rec, feats, targets, flen, target_lengths, cmask, p_torch)
    return loss, feats, flen


def decayed_names(enc, rec):
    """Weight-decay grouping of ha/optim.py:84-106 for this model: conv/linear weights and ALL
    nn.LSTM parameters (weights *and* biases, optim.py:100-101) decay; other biases do not."""
    decay, no_decay = [], []
    for prefix, d in (('encoder.', enc), ('recognizer.', rec)):
        for name in d:
            if name.startswith('lstm.'):
                decay.append(prefix + name)
            elif name.endswith('bias'):
                no_decay.append(prefix + name)
            else:
                decay.append(prefix + name)
    return sorted(decay), sorted(no_decay)


class Trainer:
    """fwd + CTC + bwd + clip(encoder only, 0.1) + AdamW  --  ha/loop.py:176-196, optim.py:132-139.

    (GradScaler is an identity on CPU: ha/loop.py:61 under a disabled autocast.)
    """

    def __init__(self, enc, rec, lr=3e-4, betas=(0.9, 0.99), weight_decay=0.01, clip=0.1):
        self.enc = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in enc.items())
        self.rec = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in rec.items())
        named = {('encoder.' + k): v for k, v in self.enc.items()}
        named.update({('recognizer.' + k): v for k, v in self.rec.items()})
        decay, no_decay = decayed_names(self.enc, self.rec)
        self.opt = torch.optim.AdamW([
            {'params': [named[n] for n in decay], 'weight_decay': weight_decay},
            {'params': [named[n] for n in no_decay], 'weight_decay': 0.0},
        ], lr=lr, betas=betas)
        self.clip = clip

    def step(self, x, input_lengths, targets, target_lengths, masks=None, p_torch=0.0):
        loss, _, _ = lstm_ctc_loss(self.enc, self.rec, x, input_lengths, targets, target_lengths, masks, p_torch)
        loss.backward()
        gnorm = torch.nn.utils.clip_grad_norm_(list(self.enc.values()), self.clip, error_if_nonfinite=False)
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return loss.detach(), gnorm.detach()


def synthetic_batch(B, T=80, F_=80, V=32, S=10, seed=42):
    """BASELINE.md section 3 inputs: randn mel frames, targets in [1,V), lengths in [5,S]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, F_, generator=g, dtype=torch.float32)
    il = torch.full((B,), T, dtype=torch.int64)
    tg = torch.randint(1, V, (B, S), generator=g, dtype=torch.int64)
    tl = torch.randint(min(5, S), S + 1, (B,), generator=g, dtype=torch.int64)
    return x, il, tg, tl
