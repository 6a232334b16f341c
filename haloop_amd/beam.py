"""Drop-in for ha/beam.py: both beam-search functions, one utterance per call like the reference,
plus a batched entry (``decode_batch``) that runs one workgroup per utterance."""
import torch

from . import _lib, ops

# ATen's CPU logaddexp sends whole chunks of 2 * Vectorized<float>::size() candidates through Sleef and the rest through libm;
# which candidates those are depends on the CPU the reference ran on.  The kernel reproduces either arithmetic bit for bit.
_CHUNK = {'AVX512': 32, 'AVX2': 16}


def set_reference_cpu(capability='AVX512'):
    """Which reference machine to reproduce: 'AVX512' (default; the fixtures under tests/golden/ were generated on one), 'AVX2',
    or any other torch.backends.cpu.get_cpu_capability() string (no vector ISA: scalar loop everywhere)."""
    _lib.check(_lib.lib().halo_set_beam_vector_chunk(_CHUNK.get(capability, 0)), 'halo_set_beam_vector_chunk')


def decode_batch(emissions, beam_size=3, log_domain=True):
    """emissions [N,T,V] on the HIP device -> (list[N] of list[beam] of list[int], scores [N,beam])."""
    N, T, V = emissions.shape
    if beam_size > 1 + V:
        raise RuntimeError('selected index k out of range')      # torch.topk at t=0, ha/beam.py:129
    seqs, lens, scores = ops.ctc_beam(emissions.float().contiguous(), beam_size, log_domain)
    seqs, lens = seqs.tolist(), lens.tolist()
    out = [[seqs[n][b][:lens[n][b]] for b in range(beam_size)] for n in range(N)]
    return out, scores


def ctc_beam_search_decode_logits(emit_logits, beam_size=3, dtype=torch.float32):
    """(T, K) log-probabilities -> (list[beam] of token lists, scores [beam]); ha/beam.py:71-137."""
    if dtype != torch.float32:
        raise NotImplementedError('the HIP beam kernel scores in float32')
    out, scores = decode_batch(emit_logits[None], beam_size, True)
    return out[0], scores[0]


def ctc_beam_search_decode_probs(emit_probs, beam_size=3):
    """(T, K) probabilities; ha/beam.py:5-68.  The reference raises NameError at beam.py:46; this
    computes what it computes once its missing ``device`` global exists (see tests/golden)."""
    out, scores = decode_batch(emit_probs[None], beam_size, False)
    return out[0], scores[0]
