"""CPU-side checks of the boundary: libhalo.so loads and exports every symbol include/halo.h
declares (no compute calls -- there is no GPU here), and the product refuses to run without it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'halo.h')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(halo_[a-z0-9_]+)\s*\(', text)))


def test_library_loads_and_exports_every_declared_symbol():
    from haloop_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    handle = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(handle, s), f'{s} declared in halo.h but not exported'
        assert s in _lib.SIGNATURES, f'{s} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == syms
    assert handle.halo_abi_version() == _lib.HALO_ABI_VERSION
    assert handle.halo_strerror(0) == b'ok'
    # size queries are host-only arithmetic and must work without a device
    H, TB = 1024, 21 * 64
    per_layer = (2 * 22 + 5 * 21) * 64 * H + 22 * 64 * H            # h, c, gates, dropped y, packed h
    fused = 4 * H * H + 8 * H * H + 21 * 64 * H                       # packed weights of both layers + packed y of layer 0
    body = (4 * H * H + 2 * per_layer + fused) * 4 + (11 + 32) * 32 * 16384
    flags = 8 * 4096                                                  # epoch words of the persistent recurrence (8 replicas, 4 KiB apart), 256-byte aligned
    # what the two-layer forward leaves for the backward: three transposed weight images; h_prev^T of both layers, an x^T slot, dropout(h0)^T
    # ... and the epoch words of the two-layer backward launch, zeroed by the forward's packing launch
    # ... and the image of the lower layer's W_ih^T (sized for an H-wide input: 8 row tiles x 128 k-tiles), also written by that launch
    packs_t = 3 * 4 * H * H * 4 + 4 * (8 * 42 * 16384) + flags + 8 * 128 * 16384
    assert handle.halo_lstm_reserve_bytes(21, 64, 128, H, 2) == (body + 255) // 256 * 256 + flags + packs_t
    assert handle.halo_lstm_status_offset(0, 21, 64, 128, H, 2) == (body + 255) // 256 * 256
    assert handle.halo_subsample_col_bytes(64, 80, 80, 5, 4, 3) == 21 * 64 * 400 * 4
    assert handle.halo_ctc_beam_workspace_bytes(1, 21, 32, 16) > 0


def test_code_object_is_gfx950_only():
    from haloop_amd import _lib
    blob = open(_lib.LIB_PATH, 'rb').read()
    targets = set(re.findall(rb'amdgcn-amd-amdhsa--(gfx[0-9a-z]+)', blob))
    assert targets == {b'gfx950'}, targets


def test_missing_library_is_fatal(monkeypatch, tmp_path):
    from haloop_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.HaloError):
        _lib.lib()


def test_modules_refuse_cpu_tensors():
    import torch
    from haloop_amd import _lib, rnn, recognizer
    enc = rnn.Encoder(12, 16, 32, num_layers=1)
    with pytest.raises(_lib.HaloError):
        enc(torch.randn(2, 20, 12), torch.tensor([20, 20]))
    rec = recognizer.TemporalClassifier(32, 9)
    with pytest.raises(_lib.HaloError):
        rec.log_probs(torch.randn(2, 5, 32))


def test_state_dict_names_match_reference_checkpoints():
    from haloop_amd import rnn, recognizer
    enc = rnn.Encoder(80, 128, 64, num_layers=2)
    assert list(enc.state_dict()) == ['subsample.weight', 'subsample.bias'] + [
        f'lstm.{n}_l{k}' for k in range(2) for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    assert rnn.Encoder().lstm.num_layers == 3 and rnn.Encoder().subsample.in_channels == 13     # ha/rnn.py:6,11
    assert list(recognizer.TemporalClassifier(64, 32).state_dict()) == ['classifier.weight', 'classifier.bias']
    dec = rnn.Decoder(50, 32, 32, 2)
    assert dec.out_layer.weight is dec.embedding.weight
    assert 'rnn.weight_hh_l1' in dec.state_dict()


def test_audio_encoder_state_dict_names_match_reference():
    """haloop_amd.attention_audio.AudioEncoder keeps ha.attention_audio.AudioEncoder's parameter names and shapes (the oracle's parameter
    dict was loaded into the reference with strict=True when the fixtures were generated)."""
    import torch
    from haloop_amd import attention, attention_audio
    from oracle import audio_encoder_ref as ae
    cfg = attention.GPTConfig(block_size=64, vocab_size=11, n_layer=2, n_head=2, n_embd=64, bias=True, causal=False, d_input=20, rotary_emb_dim=0)
    enc = attention_audio.AudioEncoder(cfg)
    want = ae.make_params(20, 64, 2, 64, True, 1)
    got = enc.state_dict()
    assert sorted(got) == sorted(want)
    assert all(tuple(got[k].shape) == tuple(want[k].shape) for k in want)
    assert torch.equal(got['transformer.wpe.weight'], want['transformer.wpe.weight'])       # the frozen sinusoid table
    assert not enc.transformer.wpe.weight.requires_grad
    assert torch.equal(enc.subsampled_lengths(torch.tensor([80, 79, 1])), ae.subsampled_lengths(torch.tensor([80, 79, 1])))
