"""Seeded synthetic inputs and parameter sets for the benchmarks and the smoke run (BASELINE.md section 3 shapes).

Product-side twins of the generators the oracle uses (oracle/cpu_ref.py, oracle/gpt_ref.py): the measured legs of bench.py and of
tools/ build their models and batches from here, so nothing under oracle/ is touched outside the CPU-baseline / parity legs.
tests/test_oracle_golden.py checks that both sides produce identical tensors.
"""
import math
from collections import OrderedDict

import torch

CONV_KERNEL = 5


def make_params(input_dim, subsample_dim, hidden_dim, num_layers, vocab_size, seed):
    """Uniform(-1/sqrt(fan), 1/sqrt(fan)) per tensor from one seeded CPU generator (state-dict keys of ha.rnn.Encoder and
    ha.recognizer.TemporalClassifier): the same weights on any box from ``seed`` alone."""
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


def synthetic_batch(B, T=80, F_=80, V=32, S=10, seed=42):
    """randn mel frames [B, T, F], full input lengths, targets in [1, V) padded to S, target lengths in [5, S]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, F_, generator=g, dtype=torch.float32)
    il = torch.full((B,), T, dtype=torch.int64)
    tg = torch.randint(1, V, (B, S), generator=g, dtype=torch.int64)
    tl = torch.randint(min(5, S), S + 1, (B,), generator=g, dtype=torch.int64)
    return x, il, tg, tl


def synthetic_tokens(B, T, vocab, seed, pad_tail=True):
    """hap-style batch (ha/score.py:57-70): targets = completions padded with 0, inputs = [eos] + completions[:-1]."""
    g = torch.Generator().manual_seed(seed)
    comp = torch.randint(1, vocab, (B, T), generator=g)
    if pad_tail:
        for b in range(B):
            n = int(torch.randint(T // 2, T + 1, (1,), generator=g))
            comp[b, n:] = 0
    eos = min(50256, vocab - 1)
    inputs = torch.cat([torch.full((B, 1), eos, dtype=torch.long), comp[:, :-1]], dim=1)
    return inputs, comp
