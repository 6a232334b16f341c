"""Drop-in for ha/recognizer.py: the CTC head (TemporalClassifier, recognizer.py:37-83) and the RNN transducer head (Transducer, :86-127)."""
import torch
import torch.nn as nn

from . import _lib, functional as HF, ops
from .ctc import ctc_reduce_mean
from .rnn import Decoder, DropoutStream
from .star import star_ctc_forward_score
from .transducer import transducer_forward_score


class TemporalClassifier(nn.Module):
    def __init__(self, feat_dim=1024, vocab_size=256):
        super().__init__()
        self.classifier = nn.Linear(feat_dim, vocab_size)
        self.dropout = nn.Dropout(0.2)
        self.dropout_stream = DropoutStream()

    def log_probs(self, features):
        if not features.is_cuda:
            raise _lib.HaloError('haloop_amd.recognizer.TemporalClassifier runs on the HIP device only')
        drop = self.dropout_stream.next(self.dropout.p, self.training)
        features = HF.dropout(features.float(), drop, _lib.HALO_STREAM_CLASSIFIER)
        logits = HF.linear(features, self.classifier.weight, self.classifier.bias)
        return HF.log_softmax(logits)

    def decode(self, features, input_lengths, target_lengths):
        # greedy, input_lengths ignored exactly like recognizer.py:48-59
        logits = self.log_probs(features)
        alignments, scores, hyp, hyp_len = ops.ctc_greedy(logits.detach().contiguous())
        lens = hyp_len.tolist()
        hypotheses = torch.nested.nested_tensor([hyp[i, :n] for i, n in enumerate(lens)])
        output_lengths = torch.tensor(lens)
        return hypotheses, output_lengths, alignments, scores, None

    def forward(self, features, targets, input_lengths=None, target_lengths=None, star_penalty=None,
                measure_entropy=False):
        if input_lengths is None:
            input_lengths = torch.full((features.shape[0],), features.shape[1], dtype=torch.long)
        if target_lengths is None:
            target_lengths = torch.full((features.shape[0],), len(targets), dtype=torch.long)
        logits = self.log_probs(features)
        if star_penalty is not None:
            # recognizer.py:74-82: the star-CTC branch reads self.star_penalty, which the reference's constructor never sets (so it raises
            # AttributeError unless the caller has assigned the attribute); same here, through the same attribute access
            dev = logits.device
            losses = star_ctc_forward_score(logits.permute(1, 0, 2), targets.to(dev), input_lengths.to(dev), target_lengths.to(dev),
                                            star_penalty=self.star_penalty)
            return ctc_reduce_mean(losses, target_lengths.to(dev)), {}
        logits1 = logits.permute(1, 0, 2)                     # T, N, C (a view; the kernel takes strides)
        dev = logits.device
        loss = HF.ctc_loss(logits1, targets.to(dev), input_lengths.to(dev), target_lengths.to(dev))
        return loss, {}


class Transducer(nn.Module):
    """RNN transducer head of the `rnn-transducer` arch (ha/recognizer.py:86-127, ha/init.py:180-185): an LSTM prediction network over
    the zero-prefixed targets, a linear transcription head on the (dropped-out) encoder features, the additive joint, and the
    transducer loss.  The reference's live branch calls torchaudio's ``rnnt_loss(joint, ..., blank=0, reduction='mean',
    fused_log_softmax=True)`` (:121-126); torchaudio is not part of this build, and the loss here is the same quantity through the
    reference's own lattice: mean over the batch of ``transducer_forward_score(joint.log_softmax(-1), ...)`` -- the equality the
    reference's tests assert (ha/transducer.py:210-231, 234-268) -- on the HIP lattice kernels, differentiable end to end."""

    def __init__(self, feat_dim=1024, vocab_size=256):
        super().__init__()
        self.classifier = nn.Linear(feat_dim, vocab_size)
        self.lm = Decoder(vocab_size, emb_dim=512, hidden_dim=512, num_layers=2, dropout=0.2)
        self.dropout = nn.Dropout(0.2)
        self.dropout_stream = DropoutStream()

    def decode(self, features, input_lengths):
        raise NotImplementedError()

    def forward(self, features, targets, input_lengths=None, target_lengths=None, star_penalty=None):   # star_penalty: ignored (:101)
        if not features.is_cuda:
            raise _lib.HaloError('haloop_amd.recognizer.Transducer runs on the HIP device only')
        dev = features.device
        N = features.shape[0]
        targets = targets.to(dev)
        hidden = self.lm.init_hidden(N)
        lm_targets = torch.cat([targets.new_zeros((N, 1)), targets], dim=1)              # input needs to start with 0 (:107)
        lm_outputs, _ = self.lm.forward_batch_first(lm_targets, hidden)                 # (N, U1, C)
        drop = self.dropout_stream.next(self.dropout.p, self.training)
        feats = HF.dropout(features.float(), drop, _lib.HALO_STREAM_CLASSIFIER)
        feats = HF.linear(feats, self.classifier.weight, self.classifier.bias)          # (N, T, C)
        joint = feats[:, :, None, :] + lm_outputs[:, None, :, :]                        # (N, T, U1, C): a broadcast add (glue)
        losses = transducer_forward_score(HF.log_softmax(joint), targets, input_lengths.to(dev), target_lengths.to(dev))
        return losses.mean(), {}
