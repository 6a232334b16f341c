"""Drop-in for the batching side of ha/symbol_tape.py (SymbolTapeNoPad, load_u16) and for get_batch of
ha/attention_loop.py:98-125, with the token tape resident in HBM: each batch is one integer-gather launch instead of a
Python loop over tape columns on the host.  Vocabularies and tokenizers are host text processing and stay out of scope."""
import math
from pathlib import Path

import numpy as np
import torch

from . import _lib
from ._lib import check, lib, ptr


def _stream():
    # the raw hipStream_t of torch's current stream: torch.cuda.current_stream() builds a Stream object per call (~4 us of
    # host time, half of what a small launch costs end to end); this is the accessor torch's own kernels launchers use
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def load_u16(filename, device='cuda'):
    """The flat little-endian u16 token file of ha/symbol_tape.py:200-204, read once and kept on the device.
    (torch has no uint16 arithmetic; like the reference's ShortTensor the values are carried in int16 storage.)"""
    data = np.fromfile(str(filename), dtype=np.int16, count=Path(filename).stat().st_size // 2)
    return torch.from_numpy(data).to(device)


class SymbolTapeNoPad:
    def __init__(self, data, batch_size, bptt_len):
        self.batch_size = batch_size
        self.bptt_len = bptt_len
        self.tape_len = math.ceil(len(data) / batch_size)
        self.tape_parts, self.trailing_tokens = divmod(self.tape_len, bptt_len)
        self.data = data
        self.pad_value = 0

    def __len__(self):
        return self.tape_parts + int(bool(self.trailing_tokens))

    def __getitem__(self, i):
        data = self.data
        if not data.is_cuda:
            raise _lib.HaloError('haloop_amd.symbol_tape.SymbolTapeNoPad batches a tape that lives on the HIP device (no CPU path)')
        if not data.is_contiguous() or data.dim() != 1:
            raise ValueError('the tape must be a flat contiguous tensor')
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        rows = self.trailing_tokens if i == self.tape_parts else self.bptt_len
        batch = torch.empty((rows, self.batch_size), dtype=data.dtype, device=data.device)
        check(lib().halo_tape_batch(ptr(data), data.element_size(), data.numel(), self.batch_size, self.bptt_len, i, rows,
                                    self.pad_value, ptr(batch), _stream()), 'halo_tape_batch')
        return batch


def get_batch(data_u16, offsets, block_size, objective='lm'):
    """data_u16: flat uint16 tokens on the device (int16 storage, see load_u16); offsets [B] int64 start positions
    -> (x, y) int64 [B, block_size] as ha/attention_loop.py:98-125 builds them ("lm": next-token targets, last column 0;
    "cond": only the final token of each row is a target)."""
    if objective not in ('lm', 'cond'):
        raise NotImplementedError(f'objective {objective!r}: only "lm" and "cond" are built')
    if not data_u16.is_cuda or data_u16.element_size() != 2:
        raise ValueError('expected a 2-byte token tape on the HIP device')
    offsets = offsets.to(device=data_u16.device, dtype=torch.int64).contiguous()
    B = offsets.numel()
    x = torch.empty(B, block_size, dtype=torch.int64, device=data_u16.device)
    y = torch.empty_like(x)
    check(lib().halo_lm_batch_u16(ptr(data_u16), data_u16.numel(), ptr(offsets), B, block_size, int(objective == 'cond'), ptr(x), ptr(y),
                                  _stream()), 'halo_lm_batch_u16')
    return x, y
