"""GPU parity of the star-CTC and transducer lattices (haloop_amd.star / haloop_amd.transducer, csrc/lattice.hip) against the
reference-generated fixtures (tests/golden/g8_star.npz, g9_transducer.npz: ha.star.star_ctc_forward_score and
ha.transducer.transducer_forward_score run on CPU, with their autograd gradients) and against the CPU oracle (oracle/star_ref.py) on
further seeded cases.  fp32 lattices: losses rel <= 1e-5 (abs 1e-4: they are sums of ~T log-probabilities), gradients <= 1e-4 rel /
5e-6 abs.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from test_oracle_golden import STAR_CASES, TRANSDUCER_CASES, star_case_from_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.fixture(scope='module')
def hal():
    from haloop_amd import _lib, star, transducer
    _lib.lib()
    return dict(star=star, transducer=transducer, lib=_lib)


@pytest.mark.parametrize('name', STAR_CASES + ['demo'])
def test_star_ctc_matches_reference(hal, name):
    g = load_golden('g8_star')
    em, tg, il, tl, pen = star_case_from_golden(g, name)
    x = em.to(DEV).requires_grad_(True)
    losses = hal['star'].star_ctc_forward_score(x, tg.to(DEV), il.to(DEV), tl.to(DEV), star_penalty=pen)
    np.testing.assert_allclose(losses.detach().cpu().numpy(), g[name + '.losses'], rtol=1e-5, atol=1e-4)
    if name == 'demo':
        return
    losses.sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[name + '.grad'], rtol=1e-4, atol=5e-6)


def test_star_ctc_weighted_backward_time_strided_and_helpers(hal):
    """Upstream gradient weights, a [N, T, C] buffer viewed time-major (what recognizer.py:77 passes), no_grad forward, and the small
    tensor helpers against the oracle's."""
    from oracle import star_ref
    star = hal['star']
    g = torch.Generator().manual_seed(11)
    T, N, C, S = 30, 6, 40, 9
    base = torch.randn(N, T, C, generator=g).log_softmax(-1)
    tg = torch.randint(1, C, (N, S), generator=g)
    il = torch.randint(20, T + 1, (N,), generator=g)
    tl = torch.randint(1, S + 1, (N,), generator=g)
    w = torch.rand(N, generator=g) + 0.5
    ref_l = star_ref.star_ctc_forward_score(base.permute(1, 0, 2), tg, il, tl, star_penalty=-1.25)
    ref_g = star_ref.star_ctc_grad(base.permute(1, 0, 2), tg, il, tl, star_penalty=-1.25) * w[None, :, None]
    x = base.to(DEV).requires_grad_(True)
    losses = star.star_ctc_forward_score(x.permute(1, 0, 2), tg.to(DEV), il.to(DEV), tl.to(DEV), star_penalty=-1.25)
    (losses * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(losses.detach().cpu().numpy(), ref_l.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(x.grad.permute(1, 0, 2).cpu().numpy(), ref_g.numpy(), rtol=1e-4, atol=5e-6)
    with torch.no_grad():
        l2 = star.star_ctc_forward_score(base.to(DEV).permute(1, 0, 2), tg.to(DEV), il.to(DEV), tl.to(DEV), star_penalty=-1.25)
    assert torch.equal(l2, losses.detach())
    s_em, s_tg = star.intersperse_stars(base.permute(1, 0, 2).to(DEV), tg.to(DEV))
    np.testing.assert_allclose(s_em.cpu().numpy(), star_ref.star_emissions(base.permute(1, 0, 2)).numpy(), rtol=1e-6, atol=1e-6)
    assert torch.equal(star.intersperse_blanks(s_tg).cpu(), star_ref.star_states(tg, C))
    with pytest.raises(NotImplementedError):
        star.star_ctc_forward_score(x.permute(1, 0, 2), tg.to(DEV), il.to(DEV), tl.to(DEV), animate=True)
    with pytest.raises(hal['lib'].HaloError):
        star.star_ctc_forward_score(base.permute(1, 0, 2), tg, il, tl)
    with pytest.raises(ValueError):
        star.star_ctc_forward_score(x.permute(1, 0, 2), tg.to(DEV), il.to(DEV) + T, tl.to(DEV))


@pytest.mark.parametrize('name', TRANSDUCER_CASES)
def test_transducer_matches_reference(hal, name):
    g = load_golden('g9_transducer')
    t = lambda k: torch.from_numpy(g[name + '.' + k])
    x = t('joint').to(DEV).requires_grad_(True)
    losses = hal['transducer'].transducer_forward_score(x, t('targets').to(DEV), t('jl').to(DEV), t('tl').to(DEV))
    np.testing.assert_allclose(losses.detach().cpu().numpy(), g[name + '.losses'], rtol=1e-5, atol=1e-4)
    losses.sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), g[name + '.grad'], rtol=1e-4, atol=5e-6)
    if name == 'batched':
        with torch.no_grad():
            l4 = hal['transducer'].transducer_forward_score4(x[0].detach(), t('targets')[0].to(DEV))
        np.testing.assert_allclose(float(l4), float(g['batched.score4_seq0']), rtol=1e-5)


def test_transducer_any_length_against_oracle(hal):
    """T = 21 (config 2's frame count): the reference's scan raises there; the recurrence it defines is checked against the oracle."""
    from oracle import star_ref
    g = torch.Generator().manual_seed(3)
    N, T, U, K = 4, 21, 6, 12
    joint = (torch.randn(N, T, 1, K, generator=g) + torch.randn(N, 1, U + 1, K, generator=g)).log_softmax(-1)
    tg = torch.randint(0, K, (N, U), generator=g)
    jl, tl = torch.tensor([21, 20, 9, 1], dtype=torch.int32), torch.tensor([6, 3, 0, 2], dtype=torch.int32)
    w = torch.rand(N, generator=g) + 0.5
    ref_l = star_ref.transducer_forward_score(joint, tg, jl, tl)
    ref_g = star_ref.transducer_grad(joint, tg, jl, tl) * w[:, None, None, None]
    x = joint.to(DEV).requires_grad_(True)
    losses = hal['transducer'].transducer_forward_score(x, tg.to(DEV), jl.to(DEV), tl.to(DEV))
    (losses * w.to(DEV)).sum().backward()
    np.testing.assert_allclose(losses.detach().cpu().numpy(), ref_l.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_g.numpy(), rtol=1e-4, atol=5e-6)
    with pytest.raises(ValueError):
        hal['transducer'].transducer_forward_score(x, tg[:, :3].to(DEV), jl.to(DEV), tl.to(DEV))


def test_temporal_classifier_star_branch(hal):
    """recognizer.py:74-82: the star-CTC branch of TemporalClassifier.forward reads self.star_penalty -- AttributeError unless the caller
    set it (the reference's constructor does not), then ctc_reduce_mean(star_ctc_forward_score(...)); gradients reach the classifier."""
    from haloop_amd import recognizer
    from oracle import star_ref
    g = torch.Generator().manual_seed(21)
    N, T, H, V, S = 5, 21, 64, 32, 6
    feats = torch.randn(N, T, H, generator=g)
    tg = torch.randint(1, V, (N, S), generator=g)
    il, tl = torch.tensor([21, 18, 21, 9, 15]), torch.tensor([6, 4, 1, 3, 5])
    rec = recognizer.TemporalClassifier(H, V).to(DEV).eval()
    with pytest.raises(AttributeError):
        rec(feats.to(DEV), tg, il, tl, star_penalty=-0.5)
    rec.star_penalty = -0.75
    loss, stats = rec(feats.to(DEV), tg, il, tl, star_penalty=-0.5)          # the argument only selects the branch, as in the reference
    loss.backward()
    w, b = rec.classifier.weight.detach().cpu().clone().requires_grad_(True), rec.classifier.bias.detach().cpu().clone().requires_grad_(True)
    em = torch.nn.functional.linear(feats, w, b).log_softmax(-1).permute(1, 0, 2)
    ref_losses = star_ref.star_ctc_forward_score(em.detach(), tg, il, tl, star_penalty=-0.75)
    ref = (ref_losses / tl).mean()
    g_em = star_ref.star_ctc_grad(em.detach(), tg, il, tl, star_penalty=-0.75) / (tl[None, :, None] * N)
    em.backward(gradient=g_em)
    assert stats == {}
    np.testing.assert_allclose(float(loss.detach()), float(ref), rtol=2e-5)
    np.testing.assert_allclose(rec.classifier.weight.grad.cpu().numpy(), w.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(rec.classifier.bias.grad.cpu().numpy(), b.grad.numpy(), rtol=2e-3, atol=2e-5)


def test_transducer_head_matches_cpu_composition(hal):
    """recognizer.Transducer (ha/recognizer.py:86-127): LSTM prediction network + linear transcription head + additive joint +
    transducer loss (mean over the batch), eval mode, against the same composition on the CPU: torch's nn.LSTM / F.linear /
    log_softmax and the pinned oracle lattice (the reference's live branch needs torchaudio's rnnt_loss, which its own tests equate
    with the lattice score).  Loss and every parameter gradient."""
    from haloop_amd import recognizer
    from oracle import star_ref
    g = torch.Generator().manual_seed(31)
    N, T, H, V, U = 3, 12, 48, 20, 5
    feats = torch.randn(N, T, H, generator=g)
    tg = torch.randint(1, V, (N, U), generator=g)
    il, tl = torch.tensor([12, 9, 12]), torch.tensor([5, 3, 4])
    torch.manual_seed(5)
    head = recognizer.Transducer(H, V).eval()
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    head = head.to(DEV)
    loss, stats = head(feats.to(DEV), tg, il, tl)
    loss.backward()
    assert stats == {}
    # CPU composition with the same parameters
    E = 512
    lstm = torch.nn.LSTM(E, E, 2)
    lstm.load_state_dict({k[len('lm.rnn.'):]: v for k, v in sd.items() if k.startswith('lm.rnn.')})
    emb = sd['lm.embedding.weight'].clone().requires_grad_(True)
    ob = sd['lm.out_layer.bias'].clone().requires_grad_(True)
    cw, cb = sd['classifier.weight'].clone().requires_grad_(True), sd['classifier.bias'].clone().requires_grad_(True)
    lm_in = torch.cat([tg.new_zeros((N, 1)), tg], dim=1)
    out, _ = lstm(torch.nn.functional.embedding(lm_in, emb).transpose(0, 1))
    lm_out = torch.nn.functional.linear(out, emb, ob).transpose(0, 1)
    f = torch.nn.functional.linear(feats, cw, cb)
    joint = (f[:, :, None, :] + lm_out[:, None, :, :]).log_softmax(-1)
    jl, tl32 = il.to(torch.int32), tl.to(torch.int32)
    ref_losses = star_ref.transducer_forward_score(joint.detach(), tg, jl, tl32)
    joint.backward(gradient=star_ref.transducer_grad(joint.detach(), tg, jl, tl32) / N)
    np.testing.assert_allclose(float(loss.detach()), float(ref_losses.mean()), rtol=2e-5)
    np.testing.assert_allclose(head.classifier.weight.grad.cpu().numpy(), cw.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(head.classifier.bias.grad.cpu().numpy(), cb.grad.numpy(), rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(head.lm.embedding.weight.grad.cpu().numpy(), emb.grad.numpy(), rtol=2e-3, atol=2e-5)
    for k, p_ in head.lm.rnn.named_parameters():
        np.testing.assert_allclose(p_.grad.cpu().numpy(), getattr(lstm, k).grad.numpy(), rtol=2e-3, atol=2e-5, err_msg=k)
    with pytest.raises(NotImplementedError):
        head.decode(feats.to(DEV), il)
