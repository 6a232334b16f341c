"""80-mel log filterbank features on the HIP device: the front-end the reference's data pipeline computes on the CPU with
``torchaudio.compliance.kaldi.fbank(wav, num_mel_bins=80)`` (ha/data.py:136-140).

``fbank(waveform, num_mel_bins=80, ...)`` keeps that function's keyword names and defaults (25 ms / 10 ms frames of 16 kHz audio, snip
edges, no dither, DC removal, pre-emphasis 0.97, povey window, power spectrum, mel filters from 20 Hz to Nyquist, natural log floored at
float32 epsilon) and returns [n_frames, num_mel_bins] float32.  Options outside that default path (dither, energy, VTLN, other windows,
``snip_edges=False``, HTK compatibility, mean subtraction) raise NotImplementedError.  Framing / DC removal / pre-emphasis / window run
in fp32 as in torchaudio (one kernel); the 512-point real DFT, |X|^2, the mel filter bank and the floored log are ONE float64 launch
against constant tables built once per configuration (csrc/fbank.hip: the arithmetic is tiny, and an fp32 DFT costs 2e-3 on the logs of
bins a few nepers below a frame's peak).

Parity: torchaudio is not part of this build (and is un-pinned in the reference's pyproject.toml), so this path is checked against
oracle/fbank_ref.py, a restatement of the published algorithm -- PARITY UNPINNED, said so there and in DESIGN.md.
"""
import math

import torch

from . import _lib, ops
from ._lib import check, lib, ptr

_CONSTANTS = {}


def _constants(num_mel_bins, frame_len, padded, sample_frequency, low_freq, high_freq, device):
    key = (num_mel_bins, frame_len, padded, float(sample_frequency), float(low_freq), float(high_freq), str(device))
    hit = _CONSTANTS.get(key)
    if hit is None:
        n = torch.arange(frame_len, dtype=torch.float64)
        window = (0.5 - 0.5 * torch.cos(2 * math.pi * n / (frame_len - 1))) ** 0.85                      # povey: symmetric hann ** 0.85
        bins = padded // 2 + 1
        nyquist = 0.5 * sample_frequency
        hi = high_freq + nyquist if high_freq <= 0.0 else high_freq
        mel = lambda f: 1127.0 * torch.log(1.0 + f / 700.0)
        mlo, mhi = mel(torch.tensor(low_freq, dtype=torch.float64)), mel(torch.tensor(hi, dtype=torch.float64))
        delta = (mhi - mlo) / (num_mel_bins + 1)
        b = torch.arange(num_mel_bins, dtype=torch.float64)[:, None]
        left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
        m = mel(sample_frequency / padded * torch.arange(padded // 2, dtype=torch.float64))[None, :]
        w = torch.clamp(torch.minimum((m - left) / (center - left), (right - m) / (right - center)), min=0.0)
        banks = torch.zeros(num_mel_bins, bins, dtype=torch.float64)
        banks[:, :padded // 2] = w                                                                       # the Nyquist column stays zero
        j = torch.arange(padded, dtype=torch.float64)
        twiddle = torch.stack([torch.cos(2 * math.pi * j / padded), torch.sin(2 * math.pi * j / padded)], dim=1).contiguous()
        hit = (window.float().to(device), twiddle.to(device), banks.contiguous().to(device), bins)
        _CONSTANTS[key] = hit
    return hit


def fbank(waveform, blackman_coeff=0.42, channel=-1, dither=0.0, energy_floor=1.0, frame_length=25.0, frame_shift=10.0, high_freq=0.0,
          htk_compat=False, low_freq=20.0, min_duration=0.0, num_mel_bins=23, preemphasis_coefficient=0.97, raw_energy=True,
          remove_dc_offset=True, round_to_power_of_two=True, sample_frequency=16000.0, snip_edges=True, subtract_mean=False,
          use_energy=False, use_log_fbank=True, use_power=True, vtln_high=-500.0, vtln_low=100.0, vtln_warp=1.0, window_type='povey'):
    """waveform [c, n] (channel ``channel``, -1 = first) or [n] -> [n_frames, num_mel_bins]."""
    if (dither != 0.0 or htk_compat or not round_to_power_of_two or not snip_edges or subtract_mean or use_energy or not use_log_fbank
            or not use_power or vtln_warp != 1.0 or window_type != 'povey'):
        raise NotImplementedError('haloop_amd.fbank builds the default path of torchaudio.compliance.kaldi.fbank only')
    if not waveform.is_cuda:
        raise _lib.HaloError('haloop_amd.fbank.fbank runs on the HIP device only (no CPU path)')
    wav = (waveform[max(channel, 0)] if waveform.dim() == 2 else waveform).float().contiguous()
    dev = wav.device
    n = wav.shape[0]
    frame_len, shift = int(sample_frequency * frame_length * 0.001), int(sample_frequency * frame_shift * 0.001)
    padded = 1 << (frame_len - 1).bit_length()
    if n < frame_len or n < min_duration * sample_frequency:
        return torch.empty(0, num_mel_bins, device=dev)
    m = 1 + (n - frame_len) // shift
    window, twiddle, banks, bins = _constants(num_mel_bins, frame_len, padded, sample_frequency, low_freq, high_freq, dev)
    s = ops._stream()
    frames = torch.empty(m, padded, device=dev)
    check(lib().halo_fbank_frames(ptr(wav), n, frame_len, shift, padded, preemphasis_coefficient, int(remove_dc_offset), ptr(window),
                                  ptr(frames), m, s), 'halo_fbank_frames')
    out = torch.empty(m, num_mel_bins, device=dev)
    # DFT, |X|^2, mel filters and the floored log of every frame in one launch, in float64 (csrc/fbank.hip)
    check(lib().halo_fbank_spectrum_mel(ptr(frames), m, padded, ptr(twiddle), ptr(banks), num_mel_bins, float(torch.finfo(torch.float32).eps),
                                        ptr(out), s), 'halo_fbank_spectrum_mel')
    return out
