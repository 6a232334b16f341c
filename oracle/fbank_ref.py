"""CPU restatement of the 80-mel filterbank front-end the reference's data pipeline applies (ha/data.py:136-140:
``torchaudio.compliance.kaldi.fbank(wav, num_mel_bins=80)``) -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: the arithmetic lives in torchaudio (a third-party dependency of the reference, un-pinned in its pyproject.toml),
which is not installed in the build container and cannot be fetched, and the reference ships no fbank fixture.  This file restates
the published algorithm of torchaudio.compliance.kaldi.fbank (itself a port of Kaldi's compute-fbank-feats) with that function's
defaults: 25 ms / 10 ms frames of 16 kHz audio, snip_edges, no dither, DC removal, pre-emphasis 0.97 (first sample against itself),
povey window (hann ** 0.85, symmetric), zero-pad to 512, power spectrum, triangular mel filters between 20 Hz and Nyquist on the
mel(f) = 1127 ln(1 + f / 700) scale, natural log floored at float32 epsilon.  numpy, float64 inside (the FFT), float32 out.
"""
import numpy as np

EPS = np.float32(np.finfo(np.float32).eps)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def povey_window(n):
    return np.hanning(n).astype(np.float64) ** 0.85            # np.hanning is the symmetric (periodic=False) Hann window


def mel_banks(num_bins, padded, sample_freq, low_freq=20.0, high_freq=0.0):
    """[num_bins, padded // 2 + 1] triangular weights (the last column, Nyquist, is zero as in kaldi's [num_bins, padded // 2])."""
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    n_fft_bins = padded // 2
    bin_width = sample_freq / padded
    mel_low, mel_high = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float64)[:, None]
    left, center, right = mel_low + b * delta, mel_low + (b + 1) * delta, mel_low + (b + 2) * delta
    mel = mel_scale(bin_width * np.arange(n_fft_bins, dtype=np.float64))[None, :]
    up, down = (mel - left) / (center - left), (right - mel) / (right - center)
    w = np.maximum(0.0, np.minimum(up, down))
    return np.concatenate([w, np.zeros((num_bins, 1))], axis=1)


def frames_of(wav, frame_len, shift):
    n = wav.shape[0]
    if n < frame_len:
        return np.zeros((0, frame_len), dtype=wav.dtype)
    m = 1 + (n - frame_len) // shift
    idx = np.arange(frame_len)[None, :] + shift * np.arange(m)[:, None]
    return wav[idx]


def fbank(wav, num_mel_bins=80, sample_frequency=16000.0, frame_length=25.0, frame_shift=10.0, preemphasis_coefficient=0.97,
          remove_dc_offset=True, low_freq=20.0, high_freq=0.0):
    wav = np.asarray(wav, dtype=np.float64).reshape(-1)
    frame_len, shift = int(sample_frequency * frame_length * 0.001), int(sample_frequency * frame_shift * 0.001)
    padded = 1 << (frame_len - 1).bit_length()
    x = frames_of(wav, frame_len, shift)
    if x.shape[0] == 0:
        return np.zeros((0, num_mel_bins), dtype=np.float32)
    if remove_dc_offset:
        x = x - x.mean(axis=1, keepdims=True)
    if preemphasis_coefficient != 0.0:
        prev = np.concatenate([x[:, :1], x[:, :-1]], axis=1)
        x = x - preemphasis_coefficient * prev
    x = x * povey_window(frame_len)[None, :]
    spec = np.fft.rfft(x, n=padded, axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ mel_banks(num_mel_bins, padded, sample_frequency, low_freq, high_freq).T
    return np.log(np.maximum(mel, EPS)).astype(np.float32)
