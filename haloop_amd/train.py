"""The LSTM-CTC training step as haloop runs it (ha/loop.py:113-196), straight on the C ABI.

One step = encoder forward -> CTC head -> CTC/classifier/LSTM/conv backward -> (data-parallel
gradient average) -> clip the ENCODER's gradient norm to 0.1 (loop.py:184) -> AdamW
(ha/optim.py:132-139: conv/linear weights and all nn.LSTM parameters decay, other biases do not).
Parameters, gradients and Adam moments live in flat fp32 buffers so clip and AdamW are one pass
each; the modules' ``nn.Parameter``s are views into them, so ``state_dict()`` stays the
reference's.  The whole step is captured in a HIP graph (~150 dependent launches otherwise).

Data parallel: one process per GPU, parameters broadcast from rank 0,
gradients averaged with one RCCL all-reduce per step over the flat gradient buffer -- the
semantics of DistributedDataParallel in ha/attention_loop.py:154.
"""
import os
import re
import time

import torch

from . import _lib, _linear, dp, ops
from .ops import Dropout, NO_DROPOUT
from .rnn import lstm_param_list


def decay_groups(encoder, recognizer):
    """(decay, no_decay) lists of (qualified name, parameter), ha/optim.py:84-106 applied to this model."""
    decay, no_decay = [], []
    for prefix, mod in (('encoder.', encoder), ('recognizer.', recognizer)):
        for name, p in mod.named_parameters():
            is_lstm = name.startswith('lstm.') or name.startswith('rnn.')
            if is_lstm or not name.endswith('bias'):
                decay.append((prefix + name, p))
            else:
                no_decay.append((prefix + name, p))
    return decay, no_decay


def _merge_adjacent(ranges):
    """Sorted (lo, hi) ranges with touching neighbours joined."""
    out = []
    for lo, hi in sorted(ranges):
        if out and out[-1][1] == lo:
            out[-1] = (out[-1][0], hi)
        else:
            out.append((lo, hi))
    return out


class FlatParams:
    """Re-homes parameters into one flat buffer laid out

        [rec no-decay | rec decay | enc no-decay | enc decay, small | enc big: lower layers' matrices | enc big: top LSTM layer's matrices]

    so that (a) the encoder (the clipped part, ha/loop.py:184) is one contiguous range, (b) each AdamW launch covers a few contiguous
    (weight-decay, clip-scale) ranges, (c) the big matrices (every LSTM ``weight_hh`` and the ``weight_ih`` of the layers above the first:
    95 % of LC-2x1024) form one contiguous block at the end, in the order their gradients become final -- the block the data-parallel
    steps exchange in pieces (``big_early``: the top layer's matrices, final first in backward; ``big_late``: the layers below) -- and
    (d) everything else (``small_range``) is one contiguous prefix."""

    def __init__(self, encoder, recognizer, pad_to=4):
        """pad_to: the buffers' length is rounded up to a multiple of it (the flat sharded data-parallel update wants equal spans)."""
        decay, no_decay = decay_groups(encoder, recognizer)
        top = getattr(encoder, 'lstm', None)
        nl = top.num_layers if top is not None else 0

        def big_layer(n):            # the LSTM layer a big matrix belongs to, or None (anything else: the small range)
            m = re.fullmatch(r'encoder\.lstm\.weight_(ih|hh)_l(\d+)', n)
            if m is None:
                return None
            layer = int(m.group(2))
            return layer if (m.group(1) == 'hh' or layer > 0) else None
        is_top = lambda n: nl > 1 and big_layer(n) == nl - 1
        enc_decay = [(n, p) for n, p in decay if n.startswith('encoder.')]
        groups = [[(n, p) for n, p in no_decay if n.startswith('recognizer.')],
                  [(n, p) for n, p in decay if n.startswith('recognizer.')],
                  [(n, p) for n, p in no_decay if n.startswith('encoder.')],
                  [(n, p) for n, p in enc_decay if big_layer(n) is None],
                  sorted([(n, p) for n, p in enc_decay if big_layer(n) is not None and not is_top(n)], key=lambda t: (big_layer(t[0]), t[0])),
                  sorted([(n, p) for n, p in enc_decay if is_top(n)], key=lambda t: t[0])]
        dev = next(encoder.parameters()).device
        off, bounds, self.slots = 0, [], []
        for gi, g in enumerate(groups):
            start = off
            for n, p in g:
                self.slots.append((n, p, off))
                off += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned
            if gi == 3:
                # the small range ends here: a multiple of 64 elements (zero padding: zero gradients, zero moments, never updated), so
                # that up to 16 ranks can cut it into equal 16-byte-aligned pieces (dp.DirectExchange's two-exchange all-reduce)
                off = (off + 63) // 64 * 64
            bounds.append((start, off))
        self.total = off
        self.encoder_range = (bounds[2][0], off)
        self.small_range = (0, bounds[4][0])
        self.big_late = bounds[4]                         # final when the whole backward has run
        self.big_early = bounds[5]                        # final after the top LSTM layer's weight-gradient launch
        self.early_range = bounds[5]                      # (the all-reduce step's first bucket)
        self.late_range = (0, bounds[5][0])
        # AdamW ranges: (begin, end, weight-decayed, clipped)
        self.ranges = [(bounds[0][0], bounds[0][1], False, False), (bounds[1][0], bounds[1][1], True, False),
                       (bounds[2][0], bounds[2][1], False, True), (bounds[3][0], off, True, True)]
        self.padded = (off + pad_to - 1) // pad_to * pad_to
        self.params = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.padded, device=dev, dtype=torch.float32)
        self.grad_views = {}
        with torch.no_grad():
            for n, p, o in self.slots:
                view = self.params[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                self.grad_views[n] = self.grads[o:o + p.numel()].view(p.shape)

    def attach_grads(self):
        for n, p, _ in self.slots:
            p.grad = self.grad_views[n]


class LstmCtcTrainer:
    DX_SLABS = int(os.environ.get('HALO_DX_SLABS', '8'))        # K-slices of the LSTM's input-gradient product that the conv backward adds while reading (ops.lstm_bwd dx_slabs)

    def __init__(self, encoder, recognizer, lr=3e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.01,
                 clip_grad_norm=0.1, seed=None, use_graph=True, process_group=None, accumulate=1, grad_dtype='f32',
                 alias_loss=False, fused_head=True, head_one_launch=True, dp_algo='rs_ag', rehearse_dp=False, gather_dtype='f32'):
        """use_graph: True -- the step replays from HIP graphs (default); False -- the same launches issued eagerly; 'auto' (one process,
        accumulate == 1) -- both are timed over the first 53 steps and the faster way stays (``auto_choice``): with the two-layer launches a
        step is 13 launches, the host enqueues them in ~0.2 ms against ~0.47 ms on the GPU, and a replay costs ~15 us more than it saves.
        accumulate: micro-batches per optimizer step (--accumulate, ha/loop.py:176-181): every step() call runs one
        forward/backward on loss / accumulate; the all-reduce, clip and AdamW run on every accumulate-th call.
        A micro-batch whose loss is NaN/Inf contributes nothing (the reference skips it, loop.py:167-174; here it still counts
        towards the cycle, because nothing synchronises with the host); an update whose gradient norm is not finite is
        skipped on the device and does not advance the Adam step count (loop.py:185-189).
        grad_dtype: wire format of the data-parallel gradient exchange -- the all-reduce, or the reduce-scatter of ``rs_ag`` --
            ('f32' = DistributedDataParallel's; 'bf16' halves its bytes on the links).
        alias_loss: step() returns ``self.loss`` itself -- ONE device scalar that every later step overwrites -- instead of a
        copy the caller owns (for loops that read each loss before the next step, or never).
        dp_algo (more than one rank): 'rs_ag' -- the sharded update cut along the step (dp.SpanSharded): the top LSTM layer's matrix
        gradients are reduce-scattered on a side stream from the moment their launch retires (beside the rest of the backward), the
        lower layers' matrices behind the backward, the small parameters are all-reduced and updated by every rank; every rank clips
        (global norm: the partials are summed over the ranks) and updates ITS chunk of each matrix span, and the chunks are
        all-gathered -- as bf16 roundings when ``gather_dtype`` allows (below).  Forward + backward are launched eagerly or replayed
        from one graph (a 2-layer stack in bf16 mode: the two-layer persistent launches); the tail is captured in a graph.
        'direct' -- the same cut and ownership as 'rs_ag' with every collective done by this library over HIP-IPC-mapped peer arenas
        (dp.DirectExchange, csrc/dp_direct.hip: each rank writes its pieces straight into the owners' buffers over the point-to-point
        links, epoch words, bounded waits) instead of by RCCL; ``process_group`` only carries the one-time exchange of IPC handles.
        'rs_ag_flat' -- round 3's form: ONE reduce-scatter / all-gather over the whole flat buffers behind the backward (also what
        ``grad_dtype='bf16'`` uses).
        gather_dtype: 'f32' (default) -- the owners' fp32 values are all-gathered: every rank's parameters (= ``state_dict()``) are
        exact after every step.  'auto' / 'bf16' -- opt-in, half the all-gather's bytes: 'auto' = bf16 when the arithmetic mode at
        construction is single-pass bf16 (every consumer of the sharded matrices multiplies by their bf16 values, so the step is the
        one an fp32 gather gives), else f32.  The other ranks' fp32 master values are then NOT in this rank's buffers between steps
        (``masters_stale``): ``trainer.state_dict()`` / ``gather_master_weights()`` exchange them (a collective: call on every rank),
        and a step taken after the arithmetic mode has changed does so by itself and continues with fp32 gathers.
        (rehearse_dp: take this path on ONE rank of an initialised process group, every collective issued over that one rank --
        how a single GPU exercises the real backend, captured graphs included.)
        'allreduce' -- DistributedDataParallel's shape: every rank averages the whole gradient (two buckets, the first overlapped
        with the lower layers' backward) and updates every parameter.  Both give the single-process step on the concatenated batch."""
        self.alias_loss = bool(alias_loss)
        self.fused_head = fused_head
        self.head_one_launch = head_one_launch
        self.encoder, self.recognizer = encoder, recognizer
        self.accumulate = int(accumulate)
        self._micro = 0
        self.betas, self.eps, self.weight_decay, self.clip = betas, eps, weight_decay, clip_grad_norm
        if dp_algo not in ('rs_ag', 'rs_ag_flat', 'allreduce', 'direct'):
            raise ValueError(f"dp_algo must be 'rs_ag', 'direct', 'rs_ag_flat' or 'allreduce', got {dp_algo!r}")
        if gather_dtype not in ('auto', 'f32', 'bf16'):
            raise ValueError(f"gather_dtype must be 'auto', 'f32' or 'bf16', got {gather_dtype!r}")
        self.world = dp.world_size(process_group)
        # gradient accumulation lives on the all-reduce path; the bf16 gradient wire format on it and on the flat sharded step
        self.dp_algo = dp_algo if ((self.world > 1 or rehearse_dp) and self.accumulate == 1) else 'allreduce'
        if self.dp_algo in ('rs_ag', 'direct') and grad_dtype == 'bf16':
            self.dp_algo = 'rs_ag_flat'
        self._rehearse_dp = bool(rehearse_dp)
        self.flat = FlatParams(encoder, recognizer, pad_to=4 * self.world if self.dp_algo != 'allreduce' else 4)
        dev = self.flat.params.device
        # the learning rate lives on the device: the optimizer launch (captured in the step graph) reads it there, so assigning
        # ``trainer.lr`` between steps -- the reference applies its schedule every step, ha/loop.py:191 -- takes effect on replay
        self._lr_dev = torch.zeros(1, device=dev, dtype=torch.float32)
        self.lr = lr
        # sticky status word of the persistent recurrences (include/halo.h): a timed-out wait sets it, the clip kernel then
        # suppresses every update, and check_status() turns it into an exception
        self.status = torch.zeros(1, device=dev, dtype=torch.int32)
        _lib.set_status_word(self.status)
        _lib.lend_scratch(device=dev)                      # split-K slabs for the under-filled GEMMs
        self.device = dev
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.counter = torch.zeros(1, device=dev, dtype=torch.int32)     # device-side step counter (dropout offset)
        self.partials = torch.zeros(_lib.HALO_SUMSQ_PARTS, device=dev, dtype=torch.float32)
        self.partials_p = torch.zeros(4 * _lib.HALO_SUMSQ_PARTS, device=dev, dtype=torch.float32)     # the producers' partials (ops.collect_grad_sumsq)
        self.coef = torch.ones(2, device=dev, dtype=torch.float32)
        self.grad_norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.loss = torch.zeros((), device=dev, dtype=torch.float32)
        self.step_count = 0                                                # step() calls that reached the optimizer
        self.adam_step = torch.zeros(1, device=dev, dtype=torch.int32)    # APPLIED updates: advanced on the device (clip_coef)
        self._ticket = torch.zeros(1, device=dev, dtype=torch.int32)      # last-workgroup ticket of the fused CTC head
        # use_graph: True -- the step replays from HIP graphs; False -- eager launches; 'auto' (one process, no accumulation) -- both are
        # timed over the first steps and the faster one stays (_auto_step; at 13 launches per step eager launches win by 2-4 % on an idle host)
        auto = use_graph == 'auto'
        if auto:
            use_graph = True
        self.use_graph = use_graph
        self._auto = {'n': 0} if (auto and self.world == 1 and not rehearse_dp and self.accumulate == 1) else None
        self.auto_choice = None
        # sharded data-parallel step in graph mode: forward + backward launched eagerly, only the tail (the collectives) replayed
        self.eager_forward_backward = os.environ.get('HALO_DP_EAGER_FB', '1') != '0'
        self.pg = process_group
        dp.broadcast_parameters(self.flat.params, process_group)          # DDP ctor semantics (C2)
        self.sharded = None
        self.masters_stale = False             # a bf16 all-gather has run since the fp32 masters were last exchanged
        self._gather_mode = None
        if self.dp_algo in ('rs_ag', 'direct'):
            f = self.flat
            bf16_gather = gather_dtype == 'bf16' or (gather_dtype == 'auto' and _lib.get_math_mode() == 'bf16')
            self._gather_mode = _lib.get_math_mode()              # the arithmetic the bf16 gather was chosen under
            try:
                if self.dp_algo == 'direct':
                    self.sharded = dp.DirectExchange(f.params, f.grads, f.big_early, f.big_late, f.small_range, process_group,
                                                     gather_bf16=bf16_gather, norm_parts=self.partials.numel())
                else:
                    self.sharded = dp.SpanSharded(f.params, f.grads, f.big_early, f.big_late, f.small_range, process_group,
                                                  always=self._rehearse_dp, gather_bf16=bf16_gather)
            except ValueError as e:       # matrix sizes that do not cut into `world` chunks of whole float4s: the flat form pads instead
                import logging
                logging.getLogger(__name__).warning('haloop_amd.train: %s; using dp_algo rs_ag_flat', e)
                self.dp_algo = 'rs_ag_flat'
            else:
                self._mid_event = torch.cuda.Event() if dev.type == 'cuda' else None
                if self._mid_event is not None:
                    self._mid_event.record()                       # (creates the handle the library records)
        if self.dp_algo == 'rs_ag_flat':
            self.sharded = dp.ShardedUpdate(self.flat.params, self.flat.grads, process_group, always=self._rehearse_dp, wire_dtype=grad_dtype)
        # two buckets in readiness order: [top layer + recognizer] then [the rest]
        self.avg_early = dp.GradientAverager(self.flat.grads, process_group, span=self.flat.early_range, wire_dtype=grad_dtype)
        self.avg_late = dp.GradientAverager(self.flat.grads, process_group, span=self.flat.late_range, wire_dtype=grad_dtype)
        self._graphs = None
        self._static = None
        self._accum = torch.zeros_like(self.flat.grads) if self.accumulate > 1 else None

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        self._lr_dev.fill_(self._lr)

    def check_status(self):
        """Raise if a persistent recurrence of any step so far gave up a bounded wait (its workgroups were not all resident: a
        CU mask, another process on the GPU, a profiler).  One device read: call it at logging intervals, not every step.
        No update has been applied since the word was raised (halo_clip_coef_step)."""
        if int(self.status.item()) != 0:
            raise _lib.HaloError('a persistent LSTM recurrence timed out waiting for its peer workgroups (they were not all resident); '
                                 'no optimizer update has been applied since.  Free the GPU of other work, or select the step-launch '
                                 'chain with haloop_amd._lib.set_lstm_persistent(False), then zero trainer.status and continue.')

    # ---- pieces -------------------------------------------------------------------------------
    def _dropout(self):
        p = self.encoder.dropout.p if self.encoder.training else 0.0
        return Dropout(p, self.seed, 0, self.counter) if p > 0 else NO_DROPOUT

    def _forward_backward(self, x, il, tg, tl):
        self._backward_rest(self._forward_backward_top(x, il, tg, tl))

    def _forward_backward_top(self, x, il, tg, tl):
        """Forward, loss, and backward down to (and including) the top LSTM layer: every gradient of
        FlatParams.early_range is final when this returns."""
        # the small fixed-order sums (the head's partials, the LSTM bias partials) ride in the conv backward's reduce launch at the end of
        # _backward_rest -- unless the first gradient bucket is all-reduced in between, which needs the head's gradients final here
        self._defer = not (self.world > 1 and self.dp_algo == 'allreduce')
        if self._defer:
            ops.defer_small_jobs.begin()
        # one process, no accumulation: the clipped gradients' squared-norm partials come from the launches that store those gradients
        self._norm_count = 0
        self._collect = self._defer and self.world == 1 and self.sharded is None and self.accumulate == 1 and self.encoder.lstm.num_layers == 2
        if self._collect:
            ops.collect_grad_sumsq(self.partials_p)
        try:
            return self._forward_backward_top_body(x, il, tg, tl)
        except BaseException:
            if self._defer:
                self._defer = False
                _lib.lib().halo_set_defer_small_jobs(0)
            if self._collect:
                self._collect = False
                _lib.lib().halo_set_grad_sumsq(None, 0)
            raise

    def _forward_backward_top_body(self, x, il, tg, tl):
        enc, rec, gv = self.encoder, self.recognizer, self.flat.grad_views
        drop = self._dropout()
        B, T, F = x.shape
        L = enc.lstm.num_layers
        lp_list = lstm_param_list(enc.lstm)
        w_ih, w_hh, b_ih, b_hh = lp_list[0::4], lp_list[1::4], lp_list[2::4], lp_list[3::4]
        H, Cc = w_hh[0].shape[1], enc.subsample.weight.shape[0]
        V = rec.classifier.weight.shape[0]
        # forward (ha/rnn.py:20-26, ha/recognizer.py:43-46,61-73)
        y_sub, col = ops.subsample_fwd(x, enc.subsample.weight, enc.subsample.bias, drop)
        Tp = y_sub.shape[0]
        feats = torch.empty(B, Tp, H, device=x.device, dtype=torch.float32)
        _, _, _, reserve = ops.lstm_fwd(y_sub, w_ih, w_hh, b_ih, b_hh, y=feats, y_strides=(H, Tp * H), y_relu=True, drop=drop)
        p_cls = rec.dropout.p if rec.training else 0.0
        cdrop = Dropout(p_cls, self.seed, 0, self.counter) if p_cls > 0 else NO_DROPOUT
        grads = {'dw_ih': [gv[f'encoder.lstm.weight_ih_l{k}'] for k in range(L)],
                 'dw_hh': [gv[f'encoder.lstm.weight_hh_l{k}'] for k in range(L)],
                 'db_ih': [gv[f'encoder.lstm.bias_ih_l{k}'] for k in range(L)],
                 'db_hh': [gv[f'encoder.lstm.bias_hh_l{k}'] for k in range(L)]}
        if self.fused_head and ops.ctc_head_supported(Tp, H, V, tg.shape[1]):
            # the whole head in three launches: dropout + Linear + log_softmax + lengths + CTC alpha + mean loss; then CTC beta +
            # log_softmax backward + d features + per-utterance d W / d b; then their fixed-order sum (csrc/head.hip)
            sid = _lib.HALO_STREAM_CLASSIFIER
            if self.head_one_launch and _lib.get_math_mode() != 'f32':
                # ... or, outside the exact-f32 mode, forward and backward in ONE launch (halo_ctc_head_train) + the deferred sum
                if getattr(self, '_head_tws', None) is None or self._head_tws_dims != (B, H, V):
                    self._head_tws, self._head_tws_dims = ops.ctc_head_train_workspace(B, H, V, x.device), (B, H, V)
                    self._head_ticket = ops.ctc_head_train_ticket(B, H, x.device)
                dfeats, _, _, _ = ops.ctc_head_train(feats, rec.classifier.weight, rec.classifier.bias, cdrop, sid, il, tg, tl, self.loss,
                                                     self._head_ticket, gv['recognizer.classifier.weight'], gv['recognizer.classifier.bias'],
                                                     workspace=self._head_tws)
                return self._lstm_backward_top(x, y_sub, col, w_ih, w_hh, reserve, grads, drop, dfeats.view(B * Tp, H),
                                               (B, T, F, Cc, H, Tp, L))
            lp, alpha, nll, flen, grad_out, (tg64, tl64) = ops.ctc_head_fwd(feats, rec.classifier.weight, rec.classifier.bias, cdrop, sid,
                                                                           il, tg, tl, self.loss, self._ticket)
            if getattr(self, '_head_ws', None) is None or self._head_ws_dims != (B, H, V):
                self._head_ws, self._head_ws_dims = ops.ctc_head_workspace(B, H, V, x.device), (B, H, V)      # outlives the deferred sum
            dfeats = ops.ctc_head_bwd(feats, rec.classifier.weight, cdrop, sid, flen, tg64, tl64, lp, alpha, nll, grad_out,
                                      gv['recognizer.classifier.weight'], gv['recognizer.classifier.bias'],
                                      workspace=self._head_ws).view(B * Tp, H)
            return self._lstm_backward_top(x, y_sub, col, w_ih, w_hh, reserve, grads, drop, dfeats, (B, T, F, Cc, H, Tp, L))
        fdrop = ops.dropout_fwd(feats, cdrop, _lib.HALO_STREAM_CLASSIFIER) if p_cls > 0 else feats
        f2d = fdrop.view(B * Tp, H)
        logits = ops.gemm(f2d, rec.classifier.weight, True, True, B * Tp, V, H, bias1=rec.classifier.bias)
        lp = ops.log_softmax_fwd(logits)
        flen, grad_out = ops.ctc_prepare(il, tl)                               # ha/rnn.py:13-18 lengths; d(mean)/d(nll)
        nll, alpha, saved = ops.ctc_fwd(lp.view(B, Tp, V), False, tg, flen, tl)
        ops.ctc_mean_loss(nll, tl, self.loss)                                  # reduction='mean', recognizer.py:71
        # backward
        dlp = ops.ctc_bwd(lp.view(B, Tp, V), False, saved, alpha, nll, grad_out)
        dlogits = ops.log_softmax_bwd(dlp.view(B * Tp, V), lp)
        ops.gemm(dlogits, f2d, False, False, V, H, B * Tp, out=gv['recognizer.classifier.weight'])
        ops.colsum(dlogits, out=gv['recognizer.classifier.bias'])
        dfeats = ops.gemm(dlogits, rec.classifier.weight, True, False, B * Tp, H, V, drop=cdrop,
                          stream_id=_lib.HALO_STREAM_CLASSIFIER)
        return self._lstm_backward_top(x, y_sub, col, w_ih, w_hh, reserve, grads, drop, dfeats, (B, T, F, Cc, H, Tp, L))

    def _lstm_backward_top(self, x, y_sub, col, w_ih, w_hh, reserve, grads, drop, dfeats, dims):
        B, T, F, Cc, H, Tp, L = dims
        ws = ops.lstm_bwd_workspace(y_sub, w_hh)
        # room for the K-slices of the input-gradient product: the conv backward adds them as it reads, no reduce launch between
        dy_sub = torch.empty((self.DX_SLABS,) + tuple(y_sub.shape), device=y_sub.device, dtype=torch.float32)
        # one process: the whole stack in one call (a 2-layer stack in bf16 mode then runs as ONE two-layer persistent launch,
        # csrc/lstm_persist2.hip); data parallel: the top layer first, so that the first gradient bucket is final early
        top = (L - 1 if L > 1 else 0) if (self.world > 1 and self.dp_algo == 'allreduce') else 0
        ops.lstm_bwd(y_sub, w_ih, w_hh, dfeats, (H, Tp * H), True, reserve, grads=grads, drop=drop, layers=(top, L),
                     workspace=ws, dx=dy_sub, dx_slabs=self.DX_SLABS)
        self._dx_slabs = ops.lstm_dx_slabs_left() if top == 0 else 1
        return (y_sub, col, w_ih, w_hh, reserve, grads, drop, ws, dy_sub, top, (B, T, F, Cc, H, Tp, L))

    def _backward_rest(self, st):
        """The lower LSTM layers and the subsample conv: completes FlatParams.late_range."""
        y_sub, col, w_ih, w_hh, reserve, grads, drop, ws, dy_sub, top, (B, T, F, Cc, H, Tp, L) = st
        gv = self.flat.grad_views
        if top > 0:
            # with data parallelism this part runs beside the first bucket's all-reduce, whose kernels hold CUs: the
            # persistent recurrence needs every workgroup resident at once, so the lower layers use the launch chain there
            # (a copy of the caller's settings with that one switch off, selected for this call only: nothing process-wide is flipped)
            if self.world > 1:
                if getattr(self, '_ctx_steps', None) is None:
                    self._ctx_steps = _lib.Context()
                    with self._ctx_steps:
                        _lib.set_lstm_persistent(False)
                with self._ctx_steps:
                    ops.lstm_bwd(y_sub, w_ih, w_hh, None, (H, Tp * H), True, reserve, grads=grads, drop=drop, layers=(0, top),
                                 workspace=ws, dx=dy_sub)
            else:
                ops.lstm_bwd(y_sub, w_ih, w_hh, None, (H, Tp * H), True, reserve, grads=grads, drop=drop, layers=(0, top),
                             workspace=ws, dx=dy_sub)
        try:
            ops.subsample_bwd(dy_sub, y_sub, col, B, T, F, Cc, drop.p, dw=gv['encoder.subsample.weight'],
                              dbias=gv['encoder.subsample.bias'], slabs=self._dx_slabs)
        finally:
            if self._defer:
                self._defer = False
                ops.defer_small_jobs.end()
            if self._collect:
                self._collect = False
                n, bits = ops.grad_sumsq_state()
                ops.collect_grad_sumsq(None)
                self._norm_count = n if bits == ops.GRAD_SUMSQ_ALL else 0      # 0: a path that did not contribute -- the plain pass runs

    def _all_reduce(self):
        self.avg_early.average()
        self.avg_late.average()

    def _warm_optimizer_kernels(self):
        """The optimizer's three kernels launched once on scratch data (nothing of the model is touched), so that their code
        objects are loaded before a stream capture records them."""
        dev = self.device
        d = [torch.zeros(8, device=dev, dtype=torch.float32) for _ in range(4)]
        parts = torch.zeros(_lib.HALO_SUMSQ_PARTS, device=dev, dtype=torch.float32)
        coef, norm = torch.ones(2, device=dev, dtype=torch.float32), torch.zeros(1, device=dev, dtype=torch.float32)
        cnt = torch.ones(1, device=dev, dtype=torch.int32)
        ops.sumsq_partials(d[1], parts)
        ops.clip_coef(parts, _lib.HALO_SUMSQ_PARTS, self.clip, coef, norm, applied_steps=cnt)
        ops.adamw_ranges(d[0], d[1], d[2], d[3], [(0, 8, 0.0, coef[0:1])], self.lr, self.betas[0], self.betas[1], self.eps, cnt)

    def _optimizer(self):
        """clip + AdamW; the update count lives on the device, so these three launches have no host scalar and are captured
        in the step graph."""
        if getattr(self, '_norm_count', 0) > 0 and self.sharded is None:
            self._apply_update(count=self._norm_count)       # the partials are already there (ops.collect_grad_sumsq)
            return
        self._norm_partials()
        self._apply_update()

    def _own_ranges(self):
        """The ranges of the flat buffers this rank's optimizer launch covers."""
        if isinstance(self.sharded, dp.SpanSharded):
            return self.sharded.own_ranges()
        if self.sharded is not None:
            return [self.sharded.span]
        return [(0, self.flat.padded)]

    def _norm_partials(self):
        """Squared-norm partials of the clipped (encoder) range -- of this rank's share of it under a sharded update (summed over the ranks
        they give the whole norm: a replicated range counts on rank 0 only)."""
        f = self.flat
        src = self.sharded.norm_ranges() if isinstance(self.sharded, dp.SpanSharded) else self._own_ranges()
        cut = [(max(f.encoder_range[0], lo), min(f.encoder_range[1], hi)) for lo, hi in src]
        cut = _merge_adjacent([(a, b) for a, b in cut if b > a])
        if not cut:
            self.partials.zero_()
        elif len(cut) == 1:
            ops.sumsq_partials(f.grads[cut[0][0]:cut[0][1]], self.partials)
        else:
            ops.sumsq_ranges(f.grads, cut, self.partials)

    def _apply_update(self, count=None):
        f = self.flat
        ops.clip_coef(self.partials_p if count else self.partials, count or _lib.HALO_SUMSQ_PARTS, self.clip, self.coef, self.grad_norm,
                      applied_steps=self.adam_step)
        # all (decay, clip) ranges (cut to what this rank updates) and the dropout step counter in one launch
        ranges = [(max(a, lo), min(b, hi), self.weight_decay if decays else 0.0, self.coef[0:1] if clipped else self.coef[1:2])
                  for a, b, decays, clipped in f.ranges for lo, hi in _merge_adjacent(self._own_ranges()) if min(b, hi) > max(a, lo)]
        if ranges:
            ops.adamw_ranges(f.params, f.grads, f.exp_avg, f.exp_avg_sq, ranges, self._lr_dev, self.betas[0], self.betas[1], self.eps,
                             self.adam_step, counter=self.counter)
        else:
            ops.counter_inc(self.counter)

    def _sharded_tail(self):
        """Behind the backward: the remaining gradient exchange -> partial norms -> their sum over the ranks -> clip + AdamW on what this
        rank owns -> all-gather.  (``_dp_marks``: profile_dp_components() brackets the pieces with events.)"""
        sh = self.sharded
        mark = self._dp_mark
        if isinstance(sh, dp.SpanSharded):
            if not self._early_started:
                sh.reduce_scatter('early')
            mark('reduce_scatter_early_tail')
            sh.reduce_scatter('late')
            mark('reduce_scatter_late')
            sh.all_reduce_small()
            mark('all_reduce_small')
        else:
            sh.reduce_scatter()
            mark('reduce_scatter')
        self._norm_partials()
        sh.all_reduce_sum(self.partials)
        mark('norm_partials_and_their_all_reduce')
        self._apply_update()
        mark('clip_and_adamw_on_owned_ranges')
        sh.all_gather()
        mark('all_gather')

    _dp_marks = None

    def _dp_mark(self, name):
        if self._dp_marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self._dp_marks.append((name, ev))

    def profile_dp_components(self, x, il, tg, tl, steps=5):
        """Sharded data-parallel step (dp_algo rs_ag / rs_ag_flat), launched EAGERLY, with a HIP event behind every piece of the step on the
        compute stream: -> {piece: average microseconds} over ``steps`` steps (after one warm-up).  ``forward_backward`` ends where the tail
        starts; ``wait_early_reduce_scatter`` is what of the side-stream reduce-scatter (started from the middle of the backward) is
        still exposed there; the rest are the tail's pieces in order.  A measurement aid (bench.py --gpus N prints it as
        config.dp_components_us): every event costs a host call, so the pieces sum to a little more than the untimed step."""
        if self.sharded is None:
            return None
        was_graph, self.use_graph = self.use_graph, False
        acc = {}
        try:
            for it in range(steps + 1):
                self._dp_marks = []
                start = torch.cuda.Event(enable_timing=True)
                start.record()
                self._early_started = False
                self.masters_stale = bool(getattr(self.sharded, 'gather_bf16', False))
                handle = self._forward_backward_overlapped(x, il, tg, tl)
                self._dp_mark('forward_backward')
                if handle is not None:
                    self.sharded.wait(handle)
                    self._early_started = True
                self._dp_mark('wait_early_reduce_scatter')
                self._sharded_tail()
                torch.cuda.synchronize()
                if it > 0:
                    prev = start
                    for name, ev in self._dp_marks:
                        acc[name] = acc.get(name, 0.0) + 1e3 * prev.elapsed_time(ev) / steps
                        prev = ev
                    acc['step_total'] = acc.get('step_total', 0.0) + 1e3 * start.elapsed_time(prev) / steps
        finally:
            self._dp_marks = None
            self.use_graph = was_graph
        return {k: round(v, 1) for k, v in acc.items()}

    def gather_master_weights(self):
        """After a bf16 all-gather (``gather_dtype``) a rank's buffers hold the OTHER ranks' matrix chunks as bf16 roundings: exchange the
        fp32 master values (one fp32 all-gather of the sharded spans) before ``state_dict()`` is saved or evaluated in another arithmetic."""
        if isinstance(self.sharded, dp.SpanSharded) and self.sharded.gather_bf16:
            self.sharded.gather_masters()
            _lib.bump_weights_epoch()
        self.masters_stale = False

    def state_dict(self):
        """{'encoder': ..., 'recognizer': ...} with every rank's fp32 master values in place (after a bf16 all-gather the other ranks'
        chunks are exchanged first: a collective, so call it on every rank) -- what a checkpoint should save (ha/loop.py:104-111)."""
        if self.masters_stale:
            self.gather_master_weights()
        return {'encoder': self.encoder.state_dict(), 'recognizer': self.recognizer.state_dict()}

    # ---- public -------------------------------------------------------------------------------
    def step(self, x, input_lengths, targets, target_lengths):
        """One optimizer step (one micro-step of it when accumulate > 1).  Returns the (device) loss of this batch: a copy the
        caller owns, or ``self.loss`` itself under ``alias_loss``; nothing here synchronises."""
        if self._auto is not None:
            loss = self._auto_step(x, input_lengths, targets, target_lengths)
        else:
            loss = self._step(x, input_lengths, targets, target_lengths)
        return loss if self.alias_loss else loss.clone()

    # use_graph='auto' (one process): steps 1-8 replay the graph (warm-up and capture), 9-28 are timed replayed, 29-33 and 34-53 run
    # as eager launches (warm-up, timed); from step 54 on the faster way runs.  Three host synchronisations in all.
    _AUTO = (8, 28, 33, 53)

    def _auto_step(self, x, il, tg, tl):
        a, (w0, g1, w1, e1) = self._auto, self._AUTO
        a['n'] += 1
        n = a['n']
        if n == w0 + 1 or n == w1 + 1:
            torch.cuda.synchronize()
            a['t0'] = time.perf_counter()
        self.use_graph = n <= g1
        loss = self._step(x, il, tg, tl)
        if n == g1 or n == e1:
            torch.cuda.synchronize()
            a['graph' if n == g1 else 'eager'] = time.perf_counter() - a['t0']
        if n == e1:
            self.use_graph = not (a['eager'] / (e1 - w1) < 0.99 * a['graph'] / (g1 - w0))      # eager must win by 1 % to be kept
            self.auto_choice = {'graph_replay_ms': 1e3 * a['graph'] / (g1 - w0), 'eager_launches_ms': 1e3 * a['eager'] / (e1 - w1),
                                'use_graph': self.use_graph}
            self._auto = None
        return loss

    def _step(self, x, input_lengths, targets, target_lengths):
        if _lib._status_tensor is not self.status:          # another trainer registered its word since: this step's launches report to OURS
            _lib.set_status_word(self.status)
        if self.accumulate > 1:
            return self._accumulating_step(x, input_lengths, targets, target_lengths)
        self.step_count += 1
        if self.sharded is not None:
            if self._gather_mode is not None and getattr(self.sharded, 'gather_bf16', False) and _lib.get_math_mode() != self._gather_mode:
                # the arithmetic changed under a bf16 gather (every rank switches together): the replicas would multiply by different
                # values.  Exchange the masters and gather in fp32 from here on (the captured tail is rebuilt).
                self.gather_master_weights()
                self.sharded.gather_bf16 = False
                self._tail_graph, self._tail_calls = None, 0
            return self._sharded_step(x, input_lengths, targets, target_lengths)
        if not self.use_graph:
            st = self._forward_backward_top(x, input_lengths, targets, target_lengths)
            w1 = self.avg_early.start()          # overlaps the rest of backward (world > 1)
            self._backward_rest(st)
            w2 = self.avg_late.start()
            self.avg_early.finish(w1)
            self.avg_late.finish(w2)
            self._optimizer()
            return self.loss
        return self._graph_step(x, input_lengths, targets, target_lengths)

    def _forward_backward_overlapped(self, x, il, tg, tl):
        """Forward + backward launched eagerly, with the reduce-scatter of the top LSTM layer's matrix gradients started on a side stream
        the moment the launch that stores them retires (halo_set_lstm_bwd_mid_event): it runs beside the lower layer's weight-gradient
        products and the front end's backward.  Returns the handle the tail waits for (None: the library did not record the event on
        this path -- not a two-layer launch -- and the tail reduces that span itself)."""
        sh = self.sharded
        ev = self._mid_event if (isinstance(sh, dp.SpanSharded) and (sh._native or getattr(sh, 'side_ok', False)) and 'early' in sh.spans) else None
        if ev is None:
            self._forward_backward(x, il, tg, tl)
            return None
        _lib.set_lstm_bwd_mid_event(ev)
        try:
            st = self._forward_backward_top(x, il, tg, tl)
            recorded = _lib.lib().halo_lstm_bwd_mid_event_recorded() == 1
        finally:
            _lib.set_lstm_bwd_mid_event(None)
        handle = sh.reduce_scatter('early', after=ev) if recorded else None
        self._backward_rest(st)
        return handle

    def _sharded_step(self, x, il, tg, tl):
        """world > 1 (or a one-rank rehearsal), dp_algo 'rs_ag' / 'rs_ag_flat'.  Graph mode: forward + backward are launched eagerly (13-16
        launches: the host keeps ahead) -- which is also what lets the first reduce-scatter start in the middle of the
        backward -- or replayed from one graph; the tail -- the remaining collectives with the optimizer pieces between them -- is captured
        in a second graph when the collective backend can be captured (RCCL can: its kernels are ordinary stream work), else it runs eagerly."""
        handle = None
        self._early_started = False
        self.masters_stale = bool(getattr(self.sharded, 'gather_bf16', False))      # this step's all-gather sends bf16 roundings
        if not self.use_graph or self.eager_forward_backward:
            handle = self._forward_backward_overlapped(x, il, tg, tl)
        else:
            self._replay_forward_backward(x, il, tg, tl)
        if handle is not None:
            self.sharded.wait(handle)                # the current stream continues behind the side stream's collective
            self._early_started = True
        if not self.use_graph:
            self._sharded_tail()
            return self.loss
        self._tail_calls = getattr(self, '_tail_calls', 0) + 1
        # (HALO_DP_CAPTURE: 1 / 0 force it; unset = capture on a one-rank group only -- the rehearsal, measured on this pool's one-GPU
        #  boxes -- and plain eager RCCL calls between real ranks: collectives inside a captured graph have never run on more than one
        #  rank here, and a capture that disagrees between ranks is a hang, not an exception)
        want_capture = os.environ.get('HALO_DP_CAPTURE', '1' if self.world == 1 else '0') != '0'
        if self._tail_calls == 2 and self.sharded._native and want_capture:
            # the first tail ran eagerly (communicator and kernels warm); record the second
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    self._sharded_tail()
                self._tail_graph, self._tail_graph_early = g, self._early_started
            except Exception as e:                       # a backend that cannot live inside a capture: eager collectives
                import logging
                logging.getLogger(__name__).warning('haloop_amd.train: capturing the collectives failed (%s: %s); they run eagerly',
                                                    type(e).__name__, e)
                torch.cuda.synchronize()
                self._tail_graph = None
        if getattr(self, '_tail_graph', None) is not None and self._tail_graph_early == self._early_started:
            _lib.bump_weights_epoch()
            self._tail_graph.replay()
        else:
            self._sharded_tail()
        return self.loss

    def _accumulating_step(self, x, il, tg, tl):
        # forward/backward writes this micro-batch's gradients into the flat buffer; they are folded into the running sum
        # scaled by 1/accumulate; the last micro-step hands the sum back and runs the (averaged, clipped) update
        if self.use_graph:
            self._replay_forward_backward(x, il, tg, tl)
        else:
            self._forward_backward(x, il, tg, tl)
        self._micro += 1
        first, last = self._micro == 1, self._micro == self.accumulate
        # alpha = 0 on the first micro-step is a plain scaled copy (never reads the old sum); a non-finite loss drops the batch
        ops.scale_add_(self._accum, self.flat.grads, 0.0 if first else 1.0, 1.0 / self.accumulate, guard=self.loss)
        if not last:
            ops.counter_inc(self.counter)                        # the next micro-batch draws fresh dropout masks
            return self.loss
        self._micro = 0
        self.flat.grads.copy_(self._accum)
        self._all_reduce()                                   # only on the last micro-step (attention_loop.py:203)
        self.step_count += 1
        self._optimizer()
        return self.loss

    def _replay_forward_backward(self, x, il, tg, tl):
        if self._graphs is None or len(self._graphs) != 1 or self._static[0].shape != x.shape or self._static[2].shape != tg.shape:
            # private buffers: refilling them for the next batch must never write into a tensor the caller still owns
            self._static = tuple(t.contiguous().clone() for t in (x, il.to(torch.int64), tg.to(torch.int64), tl.to(torch.int64)))
            sx, sil, stg, stl = self._static
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward_backward(sx, sil, stg, stl)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._forward_backward(sx, sil, stg, stl)
            self._graphs = (g,)
        for dst, src in zip(self._static, (x, il, tg, tl)):
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        _lib.bump_weights_epoch()
        self._graphs[0].replay()

    def _graph_step(self, x, il, tg, tl):
        # One graph for the whole step (the Adam update count is a device counter).  With data parallelism
        # backward is captured as TWO graphs so that the first gradient bucket's all-reduce (eager, on
        # RCCL's stream) runs beside the second graph, and the optimizer is a third.
        if self._graphs is None or self._static[0].shape != x.shape or self._static[2].shape != tg.shape:
            # private buffers: refilling them for the next batch must never write into a tensor the caller still owns
            self._static = tuple(t.contiguous().clone() for t in (x, il.to(torch.int64), tg.to(torch.int64), tl.to(torch.int64)))
            sx, sil, stg, stl = self._static
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up outside capture (lazy module loads)
                self._forward_backward(sx, sil, stg, stl)
                self._warm_optimizer_kernels()
            torch.cuda.current_stream().wait_stream(side)
            if self.world == 1:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward_backward(sx, sil, stg, stl)
                    self._optimizer()
                self._graphs = (g,)
            else:
                g1, g2, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                # the collective backend's own threads (RCCL watchdog: event queries) must not invalidate the capture, and nothing
                # of the broadcast may still be in flight
                torch.cuda.synchronize()
                mode = dict(capture_error_mode='thread_local')
                with torch.cuda.graph(g1, **mode):
                    state = self._forward_backward_top(sx, sil, stg, stl)
                with torch.cuda.graph(g2, pool=g1.pool(), **mode):
                    self._backward_rest(state)
                with torch.cuda.graph(g3, pool=g1.pool(), **mode):
                    self._optimizer()
                self._graphs = (g1, g2, g3)
                self._keep = state                       # buffers shared by the graphs stay alive
        sx, sil, stg, stl = self._static
        for dst, src in ((sx, x), (sil, il), (stg, tg), (stl, tl)):
            if src.data_ptr() != dst.data_ptr():       # a new batch: refill the captured input buffers
                dst.copy_(src)
        if len(self._graphs) == 1:
            _lib.bump_weights_epoch()
            self._graphs[0].replay()
        else:
            _lib.bump_weights_epoch()
            self._graphs[0].replay()
            w1 = self.avg_early.start()
            _lib.bump_weights_epoch()
            self._graphs[1].replay()
            w2 = self.avg_late.start()
            self.avg_early.finish(w1)
            self.avg_late.finish(w2)
            _lib.bump_weights_epoch()
            self._graphs[2].replay()
        return self.loss

    def static_inputs(self):
        """The graph's own input buffers (x, input_lengths, targets, target_lengths), available after the first step: fill
        them in place (e.g. as the destination of the host-to-device copy) and pass them to step() to avoid a device copy."""
        return self._static


class GraphedTrainStep:
    """Forward + ``loss.backward()`` of any model built from this package's modules as ONE HIP graph replay.

    The attention ASR / GPT training paths are autograd Functions over many small launches (~1,700 per `transformer:32` step at
    ~11 us of host time each, against ~12 ms of kernel time): run eagerly they are bound by the host.  ``GraphedTrainStep(forward,
    params, dropout_streams=...)`` warms ``loss = forward(*inputs)`` up on a side stream, captures forward and backward on private
    copies of the inputs, and ``step(*inputs)`` refills those copies and replays; the gradients land in the parameters' ``.grad``
    (the same tensors every replay, so an optimizer can read them in place -- keep ``set_to_none=False``).  A new input shape
    re-captures.  The weights' GEMM operand images are rebuilt by launches inside the graph, so a replay after an in-place
    optimizer step sees the new weights.

    Dropout: pass the models' ``DropoutStream`` objects; they are switched to a shared device counter that the graph itself advances
    once per replay, so every replay draws fresh Philox masks (a captured host-side offset would repeat one mask forever).
    ``forward`` must be free of host synchronisation (no ``.item()`` / python branching on device values) and must produce the same
    launch sequence for the same input shapes."""

    def __init__(self, forward, params, dropout_streams=()):
        self.forward = forward
        self.params = [p for p in params if p.requires_grad]
        self.streams = list(dropout_streams)
        dev = self.params[0].device
        self.counter = torch.zeros(1, device=dev, dtype=torch.int32)
        for s in self.streams:
            s.counter = self.counter
        self._graph = self._static = self._loss = None

    def _run(self, inputs):
        # weight operand images are part of the run (not a host-side cache hit recorded as nothing): a replay after an optimizer
        # step must multiply by the updated weights
        with _linear.graphed_run():
            loss = self.forward(*inputs)
            with torch.autograd.set_multithreading_enabled(False):     # the backward's launches come from THIS thread (the capturing one)
                loss.backward()
        if self.streams:
            ops.counter_inc(self.counter)
        return loss

    def step(self, *inputs):
        shapes = tuple((tuple(t.shape), t.dtype) for t in inputs)
        if self._graph is None or self._shapes != shapes:
            self._shapes = shapes
            self._static = tuple(t.detach().clone() for t in inputs)
            for p in self.params:
                p.grad = None
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                      # warm-up outside the capture: operand images, lazy module loads
                for _ in range(2):
                    for p in self.params:
                        p.grad = None
                    self._run(self._static)
            torch.cuda.current_stream().wait_stream(side)
            for p in self.params:
                p.grad = None                                  # the capture allocates the gradients in the graph's own pool
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._loss = self._run(self._static)
            self._graph = graph
        for dst, src in zip(self._static, inputs):
            dst.copy_(src)
        _lib.bump_weights_epoch()
        self._graph.replay()
        return self._loss.detach().clone()
