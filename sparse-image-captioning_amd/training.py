"""Native training harness — the counterpart of the step bodies in the reference's
``scripts/train_transformer.py:65-81`` and ``scripts/train_n_prune_transformer.py:132-168`` and of
``TrainingModule.compute_scst_loss`` (``utils/training.py:202-255``):

    zero_grad -> forward -> criterion -> backward -> clip_grad_value_(0.1) -> Adam(lr = Noam)       [one step]

Everything between the batch and the updated parameters is HIP on flat arenas: ``ortk_forward`` (activations kept
in the workspace), the fused criterion ``ortk_loss`` (log-probs are never materialised), ``ortk_backward`` into
the flat gradient arena, ONE RCCL all-reduce of that arena when ``torch.distributed`` is initialised (one process
per GPU, minibatch split by image; loss normalised by the GLOBAL mask sum), and ``ortk_adam_clip``.
"""
import ctypes as C
import math

import torch
import torch.distributed as dist

from . import _lib as L
from . import parallel
from .pruning import prune


def noam_rate(step, d_model, factor, warmup):
    """utils/optim.py:46-49."""
    return factor * (d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5)))


class NativeTrainer:
    def __init__(self, model, noamopt_factor=1.0, noamopt_warmup=20000, grad_clip=0.1, betas=(0.9, 0.98), eps=1e-9,
                 prune_supermask_lr=100.0, mask_eps=1e-2, sparsity_target=None, sparsity_weight=None, max_train_step=1,
                 overlap_allreduce=None, keep_grads=False, allreduce_dtype=None):
        L.require_gpu()
        self.model = model
        self.dev = model._flat.device
        assert self.dev.type == "cuda", "move the model to the GPU first"
        self.factor, self.warmup, self.clip, self.betas, self.eps = noamopt_factor, noamopt_warmup, grad_clip, betas, eps
        n = model._n_train
        self.grads = torch.zeros(n, device=self.dev)
        self.m = torch.zeros(n, device=self.dev)
        self.v = torch.zeros(n, device=self.dev)
        self.step_count = 0
        self.loss_dev = torch.zeros(1, device=self.dev)
        self.norm_dev = torch.zeros(1, device=self.dev)
        self._sum_scratch = None
        self.masked = getattr(model, "MASKED", False)
        if self.masked:
            self.train_masks = model._supermask
            self.mask_lr, self.mask_eps = prune_supermask_lr, mask_eps
            if self.train_masks:
                self.dm = torch.zeros(n, device=self.dev)
                self.mm = torch.zeros(n, device=self.dev)
                self.mv = torch.zeros(n, device=self.dev)
                # Only the ACTIVE masks are in the reference's mask-logit optimizer group (train_n_prune_transformer.py:67-82:
                # `model.active_pruning_masks()`); logits inside `prune_mask_freeze_scope` never move.  Here the group is the whole
                # mask arena, so the gradient of every frozen position is zeroed before the exchange and the update (a zero
                # gradient from step 1 on keeps Adam's moments and therefore the logit exactly unchanged).
                self.mask_active = None
                if model.mask_freeze_scope is not None:
                    act = torch.zeros(n, device=self.dev)
                    names = {k for k, _ in model.active_pruning_masks()}
                    for e in model.named_weight_entries():
                        if e["kind"] == 1 and (e["name"] + "_pruning_mask") in names:
                            act[e["offset"]:e["offset"] + e["numel"]] = 1.0
                    self.mask_active = act
            self.sparsity_target = sparsity_target
            # scripts/train_n_prune_transformer.py:306-312: weight = max(5, 1.5 / (1 - target)) unless given
            self.sparsity_weight = sparsity_weight if (sparsity_weight is not None and sparsity_weight >= 0) else (
                max(5.0, 1.5 / (1.0 - sparsity_target)) if sparsity_target is not None else 0.0)
            self.max_train_step = max_train_step
        # gradient exchange dtype: None = fp32 (the sum the reference's single process computes), "bf16" = half the bytes on the links
        assert allreduce_dtype in (None, "fp32", "bf16"), allreduce_dtype
        self.allreduce_dtype = "bf16" if allreduce_dtype == "bf16" else None
        self.world = parallel.world()
        # Data-parallel overlap: the decoder half of the backward finishes first, so its 154 MB of gradients are all-reduced
        # (RCCL, on the collective's own stream) while the encoder half still runs.  Dense models only: the masked variants
        # post-process the whole gradient arena (ortk_mask_bwd) before the exchange.  Default: on whenever world > 1.
        self.overlap = (self.world > 1 if overlap_allreduce is None else bool(overlap_allreduce)) and not self.masked
        self.valid_positions = True        # use data["cap_len"] when the batch has it (mixed precision)
        self._dec_off = int(L.lib().ortk_arena_decoder_offset(C.byref(model._ccfg)))
        self._pending = None
        # One process, dense model: the decoder half of the arena (64 % of the parameters) takes its Adam update on a second
        # stream as soon as the decoder half of the backward has finished, beside the encoder half (clip_grad_value_ is
        # element-wise, so the split is exact); the gradient arena is cleared by the update itself (ortk_adam_clip_zero).
        self.early_adam = (self.world == 1 and not self.masked and not self.overlap and self._dec_off % 4 == 0
                           and 0 < self._dec_off < self.grads.numel())      # (never beside an exchange of that half)
        # Masked models, one process: the element-wise tail of a step — straight-through mask backward, clip + Adam on the weights, clip +
        # Adam on the mask logits — as ONE pass over the arena (ortk_masked_adam_step) unless keep_grads asks for dW / dm or an all-reduce
        # has to come between the backward and Adam: configs[2] 12.33 -> 11.92 ms (scratch/masked_tail_ab.py; the same tail split over two
        # streams beside the encoder half of the backward gained nothing on top: the backward is bandwidth-bound itself).
        self.fused_masked_tail = True
        self._opt_stream = None
        self._grads_clean = False
        # keep_grads: `self.grads` still holds the step's gradient after the step (tests, diagnostics); default: the update
        # clears it on the way and the next step skips its zero-fill
        self.keep_grads = bool(keep_grads)
        # sparse training plans (enable_sparse_kernels(train=True)): the images are rebuilt from every step's mask sample; a
        # block denser than its reserved capacity would silently lose weights, so the sticky device flag is read (host sync)
        # after the first step and then every `overflow_check_every` steps, and a hit raises
        self.overflow_check_every = 50
        self.check_rollout_status = False       # scst_step: read the rollout decode's status word even with a device-side reward (host sync)

    # ------------------------------------------------------------------ pieces
    def _batch(self, data, tok_weight, rollouts=False):
        m = self.model
        feats, boxes, masks = m._prepare(data["att_feats"], data.get("boxes"), data.get("att_masks"), data.get("att_max_len"))
        # `cap_len` (host-side list / CPU tensor, one entry per caption row: decoder positions that carry a target) switches the
        # decoder to the valid positions only (ortk_batch.cap_off / row_pos); the collate function provides it
        return m._make_batch(feats, boxes, masks, data["seqs"], tok_weight, m.valid_position_tables(data) if self.valid_positions else None,
                             rollouts=rollouts)

    def encode_for_update(self, data, rows, train=False, seed=0, positions=None):
        """Phase 1 of the update pass of an SCST step, BEFORE the rollout: bf16 weight copies + encoder on `data`'s images into the
        training workspace of (B, S, rows, positions) — the shapes the update over `rows` captions of `positions` decoder positions
        (default seq_length: sampled captions) will use; `_step(..., encoded=True)` refuses any other geometry.  Returns the
        device address of the encoder memory ((B*S, d_model), the precision's activation type) for `opt["memory"]` of the rollout
        decode; the update then runs `_step(..., encoded=True)`: decoder + criterion + backward on that encoder state."""
        m, lib = self.model, L.lib()
        feats, boxes, masks = m._prepare(data["att_feats"], data.get("boxes"), data.get("att_masks"), data.get("att_max_len"))
        b = m._make_batch(feats, boxes, masks, None, None, None)
        b.R, b.T = int(rows), int(m.seq_length if positions is None else positions)
        m._sparse_plans()
        nbytes = lib.ortk_train_workspace_bytes(C.byref(m._ccfg), b.B, b.S, b.R, b.T)
        ws = m._workspace(("train", b.B, b.S, b.R, b.T), nbytes, True)
        pptr = m._eff_params_ptr(train, seed)
        L.check(lib.ortk_forward_phase(C.byref(m._ccfg), pptr, C.byref(b), L.ptr(ws), ws.numel(), None, 0, int(train), seed, 1,
                                       L.stream_ptr()), "ortk_forward_phase(1)")
        dt = C.c_int32(0)
        mem = lib.ortk_train_workspace_memory(C.byref(m._ccfg), b.B, b.S, b.R, b.T, L.ptr(ws), C.byref(dt))
        if not mem:
            raise L.OrtkError("ortk_train_workspace_memory")
        self._enc_keep = (feats, boxes, masks, ws)
        self._enc_geom = (b.B, b.S, b.R, b.T, bool(train), int(seed))
        return int(mem)

    def _fwd_bwd(self, batch, norm, train=True, seed=None, after_decoder_half=None, encoded=False):
        """forward + fused criterion + backward into self.grads; returns the seed the dropout / mask draws used.
        `after_decoder_half()`: called between the two backward phases (every gradient at offsets >= _dec_off is final)."""
        m, lib = self.model, L.lib()
        if seed is None:
            seed = m._next_seed() if train else 0
        m._sparse_plans()          # sparse training plans (enable_sparse_kernels(train=True)) ride on the config
        if encoded:
            # phase 2 runs on the workspace phase 1 carved: the same geometry (T = seq_length: sampled captions), mode and seed
            geom = (batch.B, batch.S, batch.R, batch.T, bool(train), int(seed))
            if getattr(self, "_enc_geom", None) != geom:
                raise ValueError(f"encoded=True: encode_for_update() prepared {getattr(self, '_enc_geom', None)}, the update asks for "
                                 f"{geom} (images, regions, caption rows, positions, train, seed)")
            self._enc_geom = None
        nbytes = lib.ortk_train_workspace_bytes(C.byref(m._ccfg), batch.B, batch.S, batch.R, batch.T)
        ws = m._workspace(("train", batch.B, batch.S, batch.R, batch.T), nbytes, True)
        pptr = m._eff_params_ptr(train, seed)
        # encoded: encode_for_update() has run phase 1 of this forward on this workspace (same parameters, mode and seed)
        L.check(lib.ortk_forward_phase(C.byref(m._ccfg), pptr, C.byref(batch), L.ptr(ws), ws.numel(), None, 0, int(train), seed,
                                       2 if encoded else 0, L.stream_ptr()), "ortk_forward")
        L.check(lib.ortk_loss(C.byref(m._ccfg), C.byref(batch), L.ptr(ws), ws.numel(), L.ptr(norm), L.ptr(self.loss_dev),
                              L.stream_ptr()), "ortk_loss")
        if self.overlap or after_decoder_half is not None:
            for phase in (1, 2):
                L.check(lib.ortk_backward_phase(C.byref(m._ccfg), pptr, L.ptr(self.grads), C.byref(batch), L.ptr(ws), ws.numel(),
                                                int(train), seed, phase, L.stream_ptr()), "ortk_backward_phase")
                if phase == 1 and self.overlap:      # the collective waits for the work queued so far, then runs beside phase 2
                    self._pending = parallel.allreduce_async(self.grads[self._dec_off:], dtype=self.allreduce_dtype)
                if phase == 1 and after_decoder_half is not None:
                    after_decoder_half()
        else:
            L.check(lib.ortk_backward(C.byref(m._ccfg), pptr, L.ptr(self.grads), C.byref(batch), L.ptr(ws), ws.numel(), int(train),
                                      seed, L.stream_ptr()), "ortk_backward")
        return seed

    def _allreduce(self):
        # RCCL over xGMI (SUM; every rank's loss is already divided by the GLOBAL norm): one flat 222 MB bucket, or — with the
        # overlap — the decoder half already in flight and the 80 MB encoder half now
        if self.overlap:
            parallel.allreduce_arena(self.grads[:self._dec_off], dtype=self.allreduce_dtype)
            if self._pending is not None:
                self._pending.wait()
                self._pending = None
            return
        parallel.allreduce_arena(self.grads, self.dm if (self.masked and self.train_masks) else None, dtype=self.allreduce_dtype)

    def _adam(self, p, g, m, v, lr, eps, zero=False):
        t = self.step_count
        b1, b2 = self.betas
        fn = L.lib().ortk_adam_clip_zero if zero else L.lib().ortk_adam_clip
        L.check(fn(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, b1, b2, eps, self.clip,
                   1.0 - b1 ** t, 1.0 - b2 ** t, L.stream_ptr()), "ortk_adam_clip")

    def rate(self):
        return noam_rate(max(self.step_count, 1), self.model.d_model, self.factor, self.warmup)

    # ------------------------------------------------------------------ steps
    def xe_step(self, data, train=True):
        """One XE step on ``data`` = {att_feats, boxes, att_masks, seqs, masks}; returns the loss (device scalar).
        Loss = LanguageModelCriterion(model(**data), seqs[:,1:], masks[:,1:]) (train_transformer.py:70)."""
        tok_w = data["masks"][:, 1:].contiguous().float()
        return self._step(data, tok_w, tok_w, train)

    def scst_step(self, data, reward_fn, num_samples=5, baseline="greedy", train=True, sample="random", update_dropout=False,
                  sample_dropout=None, rollout_opt=None):
        """SCST step (utils/training.py:202-255): greedy baseline + `num_samples` rollouts (no graph, cached attention), rewards
        from ``reward_fn(sample_seq (N,ns,L), greedy_seq (N,1,L) | None) -> (N*ns,)``, then ONE teacher-forced pass over
        [BOS, sample] with per-token weight mask*reward (RewardCriterion).

        ``sample``: "random" = multinomial rollouts (``scst_sample == "random"``, utils/training.py:232-237) or "beam_search" = the
        `num_samples` beams of a beam search (utils/training.py:226-231).

        Which policy is differentiated.  The reference decodes its greedy baseline under ``model.eval()``, then calls
        ``model.train()`` and draws its rollouts with every dropout on, back-propagating through those very passes
        (utils/training.py:216-237): the log-probs it differentiates are those of the dropout-perturbed policy that sampled.

        ``sample_dropout=True`` — the DEFAULT for multinomial rollouts of a training step (``sample_dropout=None`` and ``train``) —
        is that estimator: the rollouts are drawn in TRAIN mode (``ortk_decode_opts.train``, every dropout keyed by one seed) and
        the teacher-forced update pass runs with dropout under the SAME seed, so it reproduces the sampling passes' masks position
        by position (tests/test_gpu_model.py::test_train_mode_sampling_vs_oracle, ..._on_the_split_kernel).  For a dense model the
        step costs ONE encoder pass in train mode (phase 1 of the update's forward; the rollout decodes on its memory) plus the
        eval-mode encoder pass of the greedy baseline, and — where the column-split stack kernel serves the decode (mixed precision,
        <= 4 096 rows) — ONE decode: the greedy baseline rides as eval-mode rows in the launches of the train-mode rollouts.
        Elsewhere (fp32 parity mode, masked models) the baseline is its own eval-mode decode.

        ``sample_dropout=False`` is the dropout-free variant: eval-mode rollouts (greedy + samples in one decode pass) and, with
        ``update_dropout=False``, an eval-mode update — sampling policy = differentiated policy = the eval-mode model (teacher-forced
        and incremental log-probs agree to 3e-6).  NOT the reference's estimator: no dropout anywhere in the step.
        ``update_dropout=True`` recomputes under a fresh dropout pattern (rounds 1-2).

        ``rollout_opt``: extra entries for the ``opt`` dict of the ROLLOUT decodes (``temperature``, ``executor``, ...).  An extension:
        the reference's SCST path passes only ``num_random_sample`` / ``beam_size`` there (utils/training.py:226-237; it has no
        temperature option).  A greedy baseline that is a decode of its own does not take them."""
        m = self.model
        extra = dict(rollout_opt or {})
        was_training = m.training
        greedy = None
        B = data["att_feats"].size(0)
        kw = dict(att_feats=data["att_feats"], boxes=data.get("boxes"), att_masks=data.get("att_masks"), mode="sample",
                  att_max_len=data.get("att_max_len"))
        if sample_dropout is None:
            sample_dropout = bool(train) and sample == "random"
        drop_seed = None
        sparse_stream = getattr(m, "_sparse_stream", False)
        share_encoder = False
        # (decodes of earlier steps that are still unchecked stay on the model's list: check_decode_status() below reads them too)
        with torch.no_grad():
            if sample_dropout and train:
                assert sample == "random", "train-mode sampling: multinomial rollouts"
                drop_seed = m._next_seed()
                opt = dict(extra, num_random_sample=num_samples, beam_size=0, train_mode=True, drop_seed=drop_seed)
                # Dense model: the update's encoder half runs first, in TRAIN mode under drop_seed, into the training workspace; the
                # rollout's train-mode rows decode on that memory.  (Masked models draw a Bernoulli weight mask per training
                # forward while the rollout runs on the eval-mode masks: their passes share nothing.)
                share_encoder = not self.masked and not sparse_stream
                if share_encoder:
                    m.train()
                    opt["memory"] = self.encode_for_update(data, B * num_samples, train=True, seed=drop_seed)
                fused = dict(opt, with_greedy=True, sample_row_offset=parallel.sample_row_offset(B, num_samples + 1))
                if baseline == "greedy" and share_encoder and m.decode_supported(B, data["att_feats"].size(1), fused, data.get("att_max_len")):
                    seq, _ = m(**kw, opt=fused)
                    greedy, seq = seq[:, :1].contiguous(), seq[:, 1:].contiguous()
                else:
                    if baseline == "greedy":
                        m.eval()
                        # its status word is read NOW: the model keeps one decode workspace, and the rollout's launch below resets the
                        # word this decode would have raised (a no-op unless this decode ran the column-split kernel)
                        greedy, _ = m(**kw, opt=dict(beam_size=1, check_status=True))
                    seq, _ = m(**kw, opt=dict(opt, sample_row_offset=parallel.sample_row_offset(B, num_samples)))
            else:
                # One encoder pass per step: the update pass recomputes the rollout's log-probs in the SAME mode as the rollout
                # (eval-mode model) unless a dropout variant was asked for, so its encoder half runs first, into the training
                # workspace, and the rollout decodes on that memory (`opt["memory"]`); the update then only runs its decoder half.
                share_encoder = not (train and update_dropout) and not sparse_stream
                mem_opt = {}
                if share_encoder:
                    m.eval()
                    mem_opt = {"memory": self.encode_for_update(data, B * num_samples)}
                if sample == "beam_search":
                    assert num_samples > 1, "beam search needs more than one beam"
                    if baseline == "greedy":
                        greedy, _ = m(**kw, opt=dict(beam_size=1, check_status=True, **mem_opt))      # (status: as above)
                    seq, _ = m(**kw, opt=dict(extra, beam_size=num_samples, **mem_opt))
                else:
                    assert sample == "random", sample
                    # ONE decode pass for the greedy baseline and the samples (row 0 of each image is the arg-max decode):
                    # token for token what the two calls of utils/training.py:220-237 return, at half the launches
                    seq, _ = m(**kw, opt={**extra, "num_random_sample": num_samples, "beam_size": 0, "with_greedy": baseline == "greedy",
                                          # the draws are keyed by the GLOBAL row of the batch: N ranks on their shards sample what
                                          # one process samples on the whole batch
                                          "sample_row_offset": parallel.sample_row_offset(B, num_samples + (baseline == "greedy")), **mem_opt})
                    if baseline == "greedy":
                        greedy, seq = seq[:, :1].contiguous(), seq[:, 1:].contiguous()
        m.train(was_training)
        reward = reward_fn(seq, greedy)
        host_reward = not reward.is_cuda
        if host_reward or self.check_rollout_status:
            # A rollout on the column-split kernel whose exchange timed out returns all-pad captions and NaN log-probs: the update
            # would divide by a zero mask sum and Adam would write NaN into every weight.  A host-side reward has already waited for
            # the rollout, so reading its status word costs nothing there; a device-side reward_fn keeps the step free of host
            # synchronisation and relies on the decode's periodic check (first two calls, then every 64th) unless
            # `check_rollout_status` is set.
            m.check_decode_status()             # (the rollout's decode; a greedy baseline that was a call of its own has been checked)
        reward = reward.to(self.dev).float().reshape(-1)
        rows = seq.reshape(-1, seq.size(-1))
        mask = (rows != m.pad_idx).float()
        tf = dict(data)
        tf.pop("cap_len", None); tf.pop("_valid_rows", None)      # (those of the ground-truth captions)
        if host_reward and self.valid_positions:
            # The lengths of the SAMPLED captions live on the device.  A reward computed on the host (the CIDEr-D scorer of
            # scst/scorers.py) has already waited for the rollout, so reading them back costs one small copy and the update
            # pass then runs its decoder on the valid positions only (a sampled caption is tokens, EOS, then pads: its
            # weights lie in a prefix).  A device-side reward_fn keeps the whole step free of host synchronisation: padded layout.
            # (Train-mode rollouts included: every dropout site of the valid-position layout is keyed by the row's (caption,
            # position) index in the padded layout — ortk_batch.row_pos — so the update reproduces the rollout's masks there too.)
            pos = torch.arange(1, mask.size(1) + 1, device=mask.device, dtype=mask.dtype)
            tf["cap_len"] = (mask * pos).amax(1).clamp_(min=1).to(torch.int64).cpu()       # 1 + index of the last weighted position
        tf["seqs"] = torch.cat([rows.new_full((rows.size(0), 1), m.bos_idx), rows], 1)
        # rollouts=True: the teacher-forced pass masks nothing but the future — what the cached passes that drew the captions saw
        # (a sampled token may carry the PAD id; utils/training.py:252-254 drops it from the loss, later positions still attend to it)
        if drop_seed is not None:
            loss = self._step(tf, mask * reward[:, None], mask, True, seed=drop_seed, encoded=share_encoder, rollouts=True)
        else:
            loss = self._step(tf, mask * reward[:, None], mask, train and update_dropout, encoded=share_encoder, rollouts=True)
        return loss, reward, seq, greedy

    def _masked_tail(self, a, b, train, seed, coef, lr):
        """The element-wise tail of a masked step over arena range [a, b) in one pass (ortk_masked_adam_step): dW_eff -> (dW, dm)
        (straight-through, pruning/sampler.py), frozen scopes, clip + Adam on the weights (clearing the gradient), clip + Adam on the
        mask logits (their own group: train_n_prune_transformer.py:67-82)."""
        m, lib = self.model, L.lib()
        draws = m._draws(train)
        k = L.MaskedAdamArgs()
        k.w, k.g, k.mw, k.vw = m._flat[a:b].data_ptr(), self.grads[a:b].data_ptr(), self.m[a:b].data_ptr(), self.v[a:b].data_ptr()
        k.ml = m._mask_flat[a:b].data_ptr()
        if self.train_masks:
            k.mm, k.mv = self.mm[a:b].data_ptr(), self.mv[a:b].data_ptr()
            if self.mask_active is not None:
                k.active = self.mask_active[a:b].data_ptr()
        if draws is not None:
            k.draws = draws[a:b].data_ptr()
        if coef is not None:
            k.extra_coef = coef.data_ptr()
        k.n, k.index0, k.mode, k.seed = b - a, a, (1 if draws is not None else m._mode(train)), m._mask_seed(seed)
        t, (b1, b2) = self.step_count, self.betas
        k.lr_w, k.eps_w = lr, self.eps
        k.lr_m, k.eps_m = (self.mask_lr, self.mask_eps) if self.train_masks else (0.0, 1.0)
        k.beta1, k.beta2, k.clip, k.bc1, k.bc2 = b1, b2, self.clip, 1.0 - b1 ** t, 1.0 - b2 ** t
        L.check(lib.ortk_masked_adam_step(C.byref(k), L.stream_ptr()), "ortk_masked_adam_step")

    @staticmethod
    def scorer_reward_fn(scorer, refs, eos_idx=3, pad_idx=0, decode=None):
        """``reward_fn`` for :meth:`scst_step` from a :class:`..scst.CaptionScorer` and the images' reference captions:
        reward = score(sample) - score(baseline), as ``compute_scst_loss`` does (utils/training.py:239-254).  With
        ``decode`` (``tokenizer.decode``) the rows are decoded to strings and ``refs`` are the reference strings (``gts``) —
        the reference's flow, required for its word-keyed df pickles; without it ``refs`` are token-id lists and the scorer's
        document frequencies must be in token-id space (see ``CaptionScorer.score_sequences``)."""
        def fn(seq, greedy):
            sc_sample, sc_baseline = scorer.score_sequences(refs, seq, greedy, eos_idx=eos_idx, pad_idx=pad_idx, decode=decode)
            return torch.from_numpy(sc_sample - sc_baseline).float()
        return fn

    def _step(self, data, tok_weight, norm_mask, train, seed=None, encoded=False, rollouts=False):
        m = self.model
        self.step_count += 1
        if not self._grads_clean:
            self.grads.zero_()
        self._grads_clean = False
        n_norm = norm_mask.numel()
        ns = L.lib().ortk_sum_scratch_floats(n_norm)       # (0 up to 65 536 mask entries: one fixed-order launch)
        if ns > 0 and (self._sum_scratch is None or self._sum_scratch.numel() < ns):
            self._sum_scratch = torch.empty(ns, device=self.dev)
        L.check(L.lib().ortk_sum(L.ptr(norm_mask.contiguous()), n_norm, L.ptr(self._sum_scratch) if ns > 0 else None,
                                 L.ptr(self.norm_dev), L.stream_ptr()), "ortk_sum")
        parallel.reduce_scalar_sum(self.norm_dev)   # LanguageModelCriterion semantics over the GLOBAL batch
        batch = self._batch(data, tok_weight, rollouts)
        lr = self.rate()
        early = None
        if self.early_adam:
            if self._opt_stream is None:
                self._opt_stream = torch.cuda.Stream()
            d0 = self._dec_off

            def early():
                cur = torch.cuda.current_stream()
                self._opt_stream.wait_stream(cur)
                with torch.cuda.stream(self._opt_stream):
                    self._adam(m._flat[d0:m._n_train], self.grads[d0:], self.m[d0:], self.v[d0:], lr, self.eps, zero=not self.keep_grads)
        try:
            seed = self._fwd_bwd(batch, self.norm_dev, train, seed, after_decoder_half=early, encoded=encoded)
        except BaseException:
            if self._opt_stream is not None:       # a decoder-half update may already be queued: never leave it racing the caller
                torch.cuda.current_stream().wait_stream(self._opt_stream)
            raise
        loss = self.loss_dev.clone()
        parallel.reduce_scalar_sum(loss)            # every rank's partial is already divided by the GLOBAL normaliser
        if early is not None:
            # (early_adam implies a dense model: no sparse-plan overflow check can raise between the two halves of the update)
            self._adam(m._flat[:d0], self.grads[:d0], self.m[:d0], self.v[:d0], lr, self.eps, zero=not self.keep_grads)
            torch.cuda.current_stream().wait_stream(self._opt_stream)
            self._grads_clean = not self.keep_grads
            return loss
        if getattr(m, "_sparse_train", False) and (self.step_count == 1 or self.step_count % self.overflow_check_every == 0):
            m.check_sparse_overflow()
        if self.masked:
            coef = None
            one_pass = self.world == 1 and self.fused_masked_tail and not self.keep_grads      # (no exchange between the backward and Adam)
            if self.train_masks:
                if not one_pass:
                    self.dm.zero_()
                if self.sparsity_target is not None:
                    sl = m.compute_sparsity_loss(self.sparsity_target, self.sparsity_weight, self.step_count - 1,
                                                 self.max_train_step)
                    loss = loss + sl
                    coef, m._sparsity_coef = m._sparsity_coef, None
                    if self.world > 1:
                        coef = coef / self.world     # identical on every rank; the all-reduce below sums it back
            if one_pass:
                self._masked_tail(0, m._n_train, train, seed, coef, lr)
                self._grads_clean = True
                return loss
            draws = m._draws(train)
            if draws is not None:
                L.check(L.lib().ortk_mask_bwd_draws(L.ptr(self.grads), L.ptr(m._flat), L.ptr(m._mask_flat), L.ptr(draws), L.ptr(self.grads),
                                                    L.ptr(self.dm) if self.train_masks else None, m._n_train, L.ptr(coef),
                                                    L.stream_ptr()), "ortk_mask_bwd_draws")
            else:
                L.check(L.lib().ortk_mask_bwd(L.ptr(self.grads), L.ptr(m._flat), L.ptr(m._mask_flat), L.ptr(self.grads),
                                              L.ptr(self.dm) if self.train_masks else None, m._n_train, m._mode(train),
                                              m._mask_seed(seed), L.ptr(coef), L.stream_ptr()), "ortk_mask_bwd")
            if self.train_masks and self.mask_active is not None:
                self.dm.mul_(self.mask_active)
        self._allreduce()
        self._adam(m._flat[:m._n_train], self.grads, self.m, self.v, lr, self.eps, zero=not self.keep_grads)
        self._grads_clean = not self.keep_grads
        if self.masked and self.train_masks:
            # second param group of train_n_prune_transformer.py:67-82: lr = prune_supermask_lr (not Noam), eps 1e-2
            self._adam(m._mask_flat, self.dm, self.mm, self.mv, self.mask_lr, self.mask_eps)
        return loss
