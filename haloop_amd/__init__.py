"""haloop_amd -- MI355X-native engine behind haloop's acoustic / LM call surfaces.

Modules mirror the reference's: ``haloop_amd.rnn`` (ha/rnn.py), ``.recognizer`` (ha/recognizer.py), ``.ctc`` (ha/ctc.py),
``.beam`` (ha/beam.py), ``.attention`` + ``.score`` (ha/attention.py, ha/score.py: GPT scoring, training, generation),
``.transformer`` + ``.conv`` (ha/transformer.py, ha/conv.py: encoder-decoder attention ASR), ``.symbol_tape`` (token-tape
batching); ``.train`` / ``.dp`` / ``.infer`` are the fused LSTM-CTC training step, its data-parallel averaging and the
graph-captured recognizer.  All compute goes through the C ABI of ``csrc/libhalo.so`` (include/halo.h); there is no CPU
or eager fallback.
"""
from . import _lib  # noqa: F401

__all__ = ['rnn', 'recognizer', 'ctc', 'beam', 'attention', 'score', 'transformer', 'conv', 'symbol_tape', 'functional', 'ops',
           'train', 'dp', 'infer']
