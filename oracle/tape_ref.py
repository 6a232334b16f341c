"""CPU restatement of haloop's token-tape batching: SymbolTapeNoPad (ha/symbol_tape.py:239-279) and get_batch
(ha/attention_loop.py:98-125).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED by the reference itself: ``ha.symbol_tape`` and ``ha.attention_loop`` do not import in the build container
(missing third-party g2p_en / argument parsing at import), and the reference has no test or fixture for them.  The
restatement follows the source line by line and is pinned only by the values its ``__main__`` demo (symbol_tape.py:311-313:
the 48-letter alphabet tape, batch_size 2, bptt_len 8) must print, worked out by hand in tests/test_oracle_golden.py.
"""
import math

import numpy as np


class SymbolTapeNoPad:
    def __init__(self, data, batch_size, bptt_len):
        self.data = np.asarray(data)
        self.batch_size, self.bptt_len = batch_size, bptt_len
        self.tape_len = math.ceil(len(self.data) / batch_size)
        self.tape_parts, self.trailing_tokens = divmod(self.tape_len, bptt_len)
        self.pad_value = 0

    def __len__(self):
        return self.tape_parts + int(bool(self.trailing_tokens))

    def __getitem__(self, i):
        rows = self.trailing_tokens if i == self.tape_parts else self.bptt_len      # the three branches differ only in this
        batch = np.full((rows, self.batch_size), self.pad_value, dtype=self.data.dtype)
        for tape_index in range(self.batch_size):
            offset = tape_index * (self.tape_len - 1)
            part = self.data[offset + i * self.bptt_len:offset + i * self.bptt_len + rows]
            batch[:len(part), tape_index] = part
        return batch


def get_batch(data_u16, offsets, block_size, objective='lm'):
    data = np.asarray(data_u16).view(np.uint16)
    x = np.stack([data[i:i + block_size].astype(np.int64) for i in offsets])
    y = np.concatenate([x[:, 1:], np.zeros((len(x), 1), dtype=np.int64)], axis=1)
    if objective == 'cond':
        final_token = (x != 0).sum(axis=-1) - 2
        mask = np.zeros_like(y)
        mask[np.arange(len(x)), final_token] = 1
        y = y * mask
    return x, y
