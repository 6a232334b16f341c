"""Drop-in for ha/recognizer.py's CTC head (TemporalClassifier, recognizer.py:37-83)."""
import torch
import torch.nn as nn

from . import _lib, functional as HF, ops
from .ctc import ctc_reduce_mean
from .rnn import DropoutStream
from .star import star_ctc_forward_score


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
