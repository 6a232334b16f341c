"""GPU check of the 80-mel filterbank front-end (haloop_amd.fbank, ha/data.py:136-140) against oracle/fbank_ref.py -- a restatement of
torchaudio.compliance.kaldi.fbank's published algorithm -- and against g11_fbank, vectors of the Hugging Face transformers port of that
function (torchaudio itself is not installable here; see the oracle's header).
Log-mel features <= 1e-4 abs (framing in fp32 as torchaudio's; the DFT, the power spectrum, the filters and the log in float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,seed', [(16000, 0), (400, 1), (12345, 2), (399, 3)])
def test_fbank_matches_restatement(n, seed):
    from haloop_amd import fbank
    from oracle import fbank_ref
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(n) / 16000.0
    wav = 0.3 * torch.sin(2 * np.pi * 440.0 * t) + 0.05 * torch.randn(n, generator=g) + 0.01
    got = fbank.fbank(wav[None].cuda(), num_mel_bins=80)
    ref = fbank_ref.fbank(wav.numpy(), num_mel_bins=80)
    assert got.shape == ref.shape and got.dtype == torch.float32
    if n >= 400:
        assert got.shape == (1 + (n - 400) // 160, 80)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-4)
        if n == 16000:        # the 440 Hz tone: the loudest filter is the one whose triangle covers 440 Hz
            centers = 700.0 * (np.exp((fbank_ref.mel_scale(20.0) + (np.arange(80) + 1) * (fbank_ref.mel_scale(8000.0) - fbank_ref.mel_scale(20.0)) / 81) / 1127.0) - 1.0)
            assert abs(centers[int(got.mean(0).argmax())] - 440.0) < 40.0


def test_fbank_matches_the_transformers_port_vectors():
    from haloop_amd import fbank
    from conftest import load_golden
    g = load_golden('g11_fbank')
    for k in ('tone', 'noise', 'chirp', 'one_frame'):
        got = fbank.fbank(torch.from_numpy(g[f'{k}.wav'])[None].cuda(), num_mel_bins=80)
        ref = g[f'{k}.fbank'].astype(np.float64)
        assert tuple(got.shape) == ref.shape
        got = got.cpu().numpy().astype(np.float64)
        # framing (DC removal, pre-emphasis, window) runs in fp32 here and in float64 in the port; everything behind it in float64 on
        # both sides: the logs agree to 1e-4 wherever the energy is within 12 nepers (a factor 1.6e5) of the frame's peak, and the
        # energies to fp32 rounding of the peak everywhere (the noiseless chirp spans e^35)
        peak = np.exp(ref).max(axis=1, keepdims=True)
        assert (np.abs(np.exp(got) - np.exp(ref)) / peak).max() <= 5e-6, k          # (a float32 log near 23 resolves 2e-6 of its energy)
        near = ref >= ref.max(axis=1, keepdims=True) - 12.0
        assert np.abs(got - ref)[near].max() <= 1e-4, k


def test_fbank_refusals():
    from haloop_amd import _lib, fbank
    wav = torch.zeros(1, 1600)
    with pytest.raises(_lib.HaloError):
        fbank.fbank(wav, num_mel_bins=80)
    with pytest.raises(NotImplementedError):
        fbank.fbank(wav.cuda(), num_mel_bins=80, dither=1.0)
    assert fbank.fbank(wav.cuda(), num_mel_bins=23).shape == (8, 23)
