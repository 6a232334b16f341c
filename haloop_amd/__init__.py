"""haloop_amd -- MI355X-native engine behind haloop's acoustic call surfaces.

Modules mirror the reference's: ``haloop_amd.rnn`` (ha/rnn.py), ``haloop_amd.recognizer``
(ha/recognizer.py), ``haloop_amd.ctc`` (ha/ctc.py), ``haloop_amd.beam`` (ha/beam.py).  All compute
goes through the C ABI of ``csrc/libhalo.so`` (include/halo.h); there is no CPU or eager fallback.
"""
from . import _lib  # noqa: F401

__all__ = ['rnn', 'recognizer', 'ctc', 'beam', 'functional', 'ops', 'train']
