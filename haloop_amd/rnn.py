"""Drop-in for ha/rnn.py: LSTM acoustic Encoder (rnn.py:5-26) and LSTM LM Decoder (rnn.py:30-77).

Same constructor arguments, attribute/state-dict names and return tuples; the compute runs in
libhalo (conv-as-GEMM subsample, step-fused LSTM).  Parameters stay ordinary ``nn.Parameter``s so
checkpoints written by the reference load unchanged.
"""
import math

import torch
import torch.nn as nn

from . import _lib, functional as HF
from .ops import Dropout, NO_DROPOUT


class LSTMParams(nn.Module):
    """Parameter container with nn.LSTM's names, shapes and default init (no compute of its own)."""

    def __init__(self, input_size, hidden_size, num_layers=1, dropout=0.0, batch_first=False):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.dropout, self.batch_first = float(dropout), batch_first
        bound = 1.0 / math.sqrt(hidden_size)
        for k in range(num_layers):
            in_dim = input_size if k == 0 else hidden_size
            for name, shape in ((f'weight_ih_l{k}', (4 * hidden_size, in_dim)), (f'weight_hh_l{k}', (4 * hidden_size, hidden_size)),
                                (f'bias_ih_l{k}', (4 * hidden_size,)), (f'bias_hh_l{k}', (4 * hidden_size,))):
                self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-bound, bound)))

    def extra_repr(self):
        return f'{self.input_size}, {self.hidden_size}, num_layers={self.num_layers}, dropout={self.dropout}'


def lstm_param_list(lstm):
    """[w_ih, w_hh, b_ih, b_hh] * L from an LSTMParams or an nn.LSTM (callers may swap ``.lstm``)."""
    out = []
    for k in range(lstm.num_layers):
        out += [getattr(lstm, f'weight_ih_l{k}'), getattr(lstm, f'weight_hh_l{k}'),
                getattr(lstm, f'bias_ih_l{k}'), getattr(lstm, f'bias_hh_l{k}')]
    return out


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


class DropoutStream:
    """Host-side bookkeeping of the Philox stream: one (seed, offset) per training forward.

    Without an explicit seed, every instance gets its own key derived from torch's initial seed and the order of construction
    (splitmix64), so that two modules whose dropout sites carry the same stream ids (an encoder and a decoder, two GPTs in one
    process) draw independent masks, as torch's global generator would give them; runs with the same torch seed and the same
    construction order repeat.  Tests and the oracle pin ``.seed`` explicitly."""
    _instances = 0

    def __init__(self, seed=None):
        if seed is None:
            seed = _splitmix64((torch.initial_seed() & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(DropoutStream._instances))
            DropoutStream._instances += 1
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.offset = 0
        self.counter = None      # optional device uint32 advanced inside captured graphs

    def next(self, p, training):
        if not training or p <= 0.0:
            return NO_DROPOUT
        d = Dropout(p, self.seed, self.offset, self.counter)
        if self.counter is None:
            self.offset += 1
        return d


class Encoder(nn.Module):
    def __init__(self, input_dim=13, subsample_dim=128, hidden_dim=1024, num_layers=3):
        super().__init__()
        self.dropout = nn.Dropout(0.2)
        self.subsample = nn.Conv1d(input_dim, subsample_dim, kernel_size=5, stride=4, padding=3)
        self.lstm = LSTMParams(subsample_dim, hidden_dim, num_layers=num_layers, dropout=0.2, batch_first=True)
        self.dropout_stream = DropoutStream()

    def subsampled_lengths(self, input_lengths):
        # float floor like ha/rnn.py:13-18, result int32
        p, k, s = self.subsample.padding[0], self.subsample.kernel_size[0], self.subsample.stride[0]
        o = input_lengths + 2 * p - k
        return torch.floor(o / s + 1).int()

    def forward(self, inputs, input_lengths, measure_entropy=False):
        if not inputs.is_cuda:
            raise _lib.HaloError('haloop_amd.rnn.Encoder runs on the HIP device only (no CPU path)')
        p_conv, p_lstm = self.dropout.p, float(self.lstm.dropout)
        if self.training and p_conv != p_lstm and p_conv > 0 and p_lstm > 0 and self.lstm.num_layers > 1:
            raise NotImplementedError('subsample and inter-layer dropout rates must agree (both 0.2 in ha/rnn.py)')
        drop = self.dropout_stream.next(max(p_conv, p_lstm), self.training)
        feats = HF.encoder_forward(inputs.float(), self.subsample.weight, self.subsample.bias,
                                   lstm_param_list(self.lstm), self.lstm.num_layers, drop)
        return feats, self.subsampled_lengths(input_lengths), {}


class Decoder(nn.Module):
    def __init__(self, vocab_size, emb_dim, hidden_dim, num_layers, dropout=0.0):
        super().__init__()
        self.num_classes = vocab_size
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.embedding = nn.Embedding(vocab_size, emb_dim)
        self.rnn = LSTMParams(emb_dim, hidden_dim, num_layers, dropout=dropout)
        self.out_layer = nn.Linear(hidden_dim, vocab_size)
        self.out_layer.weight = self.embedding.weight       # tied (requires emb_dim == hidden_dim), rnn.py:42
        self.dropout_stream = DropoutStream()

    def _run(self, emb_tm, state):
        drop = self.dropout_stream.next(float(self.rnn.dropout), self.training)
        y, hn, cn = HF.lstm_forward(emb_tm, state, lstm_param_list(self.rnn), self.num_layers, drop)
        return HF.linear(y, self.out_layer.weight, self.out_layer.bias), (hn, cn)

    def forward(self, input, state):
        emb = HF.embedding(input, self.embedding.weight)     # (T, N, E)
        output, state = self._run(emb, state)
        return output.view(-1, self.num_classes), state

    def forward_batch_first(self, input, state):
        emb = HF.embedding(input, self.embedding.weight).transpose(0, 1)   # (T, N, E)
        output, state = self._run(emb, state)
        return output.transpose(0, 1), state

    def init_hidden(self, batch_size=1):
        weight = self.out_layer.weight
        h = weight.new_zeros(self.num_layers, batch_size, self.hidden_dim)
        c = weight.new_zeros(self.num_layers, batch_size, self.hidden_dim)
        return (h, c)

    def truncate_hidden(self, state):
        h, c = state
        return (h.detach(), c.detach())
