"""Object Relation Transformer on MI355X — host-side mirror of the reference's
``sparse_caption/models/relation_transformer.py:297-426`` (+ ``caption_model.py:24-28``,
``transformer.py:417-561``).  All arithmetic happens in libortk.so; this class owns

* the flat fp32 parameter arena (every ``nn.Parameter`` is a view into it, with the reference's state_dict key
  names, so ``state_dict()`` / ``load_state_dict(strict=True)`` are interchangeable with the reference),
* the drop-in call contract: ``model(att_feats, boxes, seqs, att_masks)`` -> log-probs ``(R, T, V)`` that
  participate in torch autograd, ``model(..., mode="sample", opt=...)`` -> ``(seq, seq_logprobs)``.
"""
import ctypes as C
import os
import math
import weakref

import torch
import torch.nn as nn

from . import register_model
from .. import _lib as L
from ..data.collate import ObjectRelationCollate


class _Node(nn.Module):
    """Structural placeholder so that parameter paths equal the reference's module paths."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("structural node: compute happens in RelationTransformerModel")


def parse_share_layer(value, n_layers, what):
    """``share_layer_encoder / share_layer_decoder`` (transformer.py:599-614: a sequence of layer ids such as (0, 0, 1, 1, 2, 2))
    -> per position, 0 = first use of that id, k > 0 = the same module as position k - 1."""
    if value is None or value == "" or value == ():
        return [0] * n_layers
    if isinstance(value, str):
        value = [int(v) for v in value.replace("(", "").replace(")", "").replace("[", "").replace("]", "").split(",") if v.strip()]
    if not isinstance(value, (tuple, list)):
        raise TypeError(f"`{what}` must be a tuple or list, saw {type(value)}")          # relation_transformer.py:83-84
    value = [int(v) for v in value]
    if len(value) != n_layers:
        raise NotImplementedError(f"`{what}` has {len(value)} entries but num_layers = {n_layers}: the HIP path keeps one layer "
                                  "count for both stacks")
    if sorted(set(value)) != list(range(len(set(value)))):
        raise IndexError(f"`{what}` ids must be 0..{len(set(value)) - 1}")              # relation_transformer.py:85-86 indexes a list
    first, out = {}, []
    for l, v in enumerate(value):
        out.append(0 if v not in first else first[v] + 1)
        first.setdefault(v, l)
    return out


def parse_share_att(value, what):
    """``share_att_encoder / share_att_decoder`` (relation_transformer.py:140, transformer.py:223: None, "kv" or "qk")
    -> ortk_config.share_att_* (0, 1, 2)."""
    if value in (None, "", "None", "none"):          # str_to_none, transformer.py:591
        return 0
    assert value in ("kv", "qk"), f"Invalid `share_att`: {value}"           # the reference's own assertion
    return 1 if value == "kv" else 2


def make_ccfg(config, precision, drop, train_drop_src=None, no_box=False):
    c = L.Config()
    c.no_box = 1 if no_box else 0
    get = config.get if hasattr(config, "get") else (lambda k, d=None: getattr(config, k, d))
    c.share_att_enc = parse_share_att(get("share_att_encoder", None), "share_att_encoder")
    c.share_att_dec = parse_share_att(get("share_att_decoder", None), "share_att_decoder")
    for l, v in enumerate(parse_share_layer(get("share_layer_encoder", None), int(config.num_layers), "share_layer_encoder")):
        c.share_enc[l] = v
    for l, v in enumerate(parse_share_layer(get("share_layer_decoder", None), int(config.num_layers), "share_layer_decoder")):
        c.share_dec[l] = v
    c.d_model, c.d_ff = int(config.d_model), int(config.dim_feedforward)
    c.n_layers, c.n_heads = int(config.num_layers), int(config.num_heads)
    c.vocab, c.feat, c.seq_len = int(config.vocab_size), int(config.att_feat_size), int(config.max_seq_length)
    c.pad_id, c.bos_id, c.eos_id, c.unk_id = (int(config.pad_token_id), int(config.bos_token_id),
                                              int(config.eos_token_id), int(config.unk_token_id))
    c.box_trig = 0 if get("no_box_trigonometric_embedding", False) else 1
    c.precision = int(precision)
    c.drop_src = float(config.drop_prob_src if train_drop_src is None else train_drop_src)
    c.drop = float(drop)
    return c


def arena_entries(ccfg):
    lib = L.lib()
    n = lib.ortk_arena_entries(C.byref(ccfg))
    if n < 0:
        raise L.OrtkError("unsupported model geometry (see ortk_model.hip: check_cfg)")
    out = []
    name = C.create_string_buffer(128)
    off, numel, ndim, kind = C.c_int64(), C.c_int64(), C.c_int32(), C.c_int32()
    shape = (C.c_int64 * 4)()
    for i in range(n):
        L.check(lib.ortk_arena_entry(C.byref(ccfg), i, name, C.byref(off), C.byref(numel), C.byref(ndim), shape,
                                     C.byref(kind)), "ortk_arena_entry")
        out.append(dict(name=name.value.decode(), offset=off.value, numel=numel.value,
                        shape=tuple(shape[j] for j in range(ndim.value)), kind=kind.value))
    return out


class _ForwardFn(torch.autograd.Function):
    """Teacher-forced log-probs with the backward routed through ortk_backward."""

    @staticmethod
    def forward(ctx, model, batch, train, seed, *params):
        logp, ws = model._run_forward(batch, train, seed, want_logp=True, cache_ws=False)
        ctx.model_ref = weakref.ref(model)
        ctx.batch, ctx.ws, ctx.train, ctx.seed = batch, ws, train, seed
        ctx.save_for_backward(logp)
        return logp

    @staticmethod
    def backward(ctx, dlogp):
        model = ctx.model_ref()
        (logp,) = ctx.saved_tensors
        grads = model._run_backward_external(ctx.batch, ctx.ws, logp, dlogp.contiguous(), ctx.train, ctx.seed)
        ctx.ws = None
        return (None, None, None, None) + tuple(grads)


class CaptionModelBase(nn.Module):
    """``CaptionModel.forward`` mode dispatch (caption_model.py:24-28)."""

    def forward(self, *args, **kwargs):
        mode = kwargs.pop("mode", "forward")
        return getattr(self, "_" + mode)(*args, **kwargs)


@register_model("relation_transformer")
class RelationTransformerModel(CaptionModelBase):
    COLLATE_FN = ObjectRelationCollate     # what the caller builds its data loaders from (utils/training.py:78-81)
    DROPOUT = 0.1          # make_model(dropout=0.1), relation_transformer.py:306
    MASKED = False
    NO_BOX = False         # True in the plain `transformer` subclass (models/transformer.py)

    def __init__(self, config, precision=None):
        super().__init__()
        self.config = config
        # attributes the callers read (transformer.py:418-437; utils/training.py:253)
        self.d_model, self.dim_feedforward = config.d_model, config.dim_feedforward
        self.num_layers, self.num_heads = config.num_layers, config.num_heads
        self.drop_prob_src = config.drop_prob_src
        self.seq_length = config.max_seq_length
        self.att_feat_size, self.vocab_size = config.att_feat_size, config.vocab_size
        self.eos_idx, self.bos_idx = config.eos_token_id, config.bos_token_id
        self.unk_idx, self.pad_idx = config.unk_token_id, config.pad_token_id
        self.box_trigonometric_embedding = not (config.get("no_box_trigonometric_embedding", False) if hasattr(config, "get")
                                                else getattr(config, "no_box_trigonometric_embedding", False))
        assert self.num_layers > 0, "num_layers should be greater than 0"
        if precision is None:
            precision = config.get("ortk_precision", 0) if hasattr(config, "get") else 0
        self.precision = {"fp32": 0, "f32": 0, "bf16": 1}.get(precision, precision)
        self._ccfg = make_ccfg(config, self.precision, self.DROPOUT, no_box=self.NO_BOX)
        self._entries = arena_entries(self._ccfg)
        lib = L.lib()
        self._n_train = lib.ortk_arena_numel(C.byref(self._ccfg))
        self._n_all = lib.ortk_arena_numel_with_buffers(C.byref(self._ccfg))
        self._flat = torch.zeros(self._n_all)
        self._params = {}
        self._build_tree()
        self._bind()
        self.reset_parameters()
        self._ws_cache = {}
        self._vr_stage = {}                # pinned staging rings of the valid-position tables, per (caption rows, positions)
        self._seed_counter = 0
        self.done_beams = None

    # ------------------------------------------------------------------ arena plumbing
    def _extra_param_specs(self, entry):
        """(suffix, arena_attr) of additional per-entry parameters (the `_prune` variant adds masks)."""
        return []

    def _build_tree(self):
        for e in self._entries:
            parts = e["name"].split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            if e["kind"] == 2:
                node.register_buffer(parts[-1], torch.empty(0))
                self._params[e["name"]] = (node, parts[-1], None)
            else:
                par = nn.Parameter(torch.empty(0))
                node.register_parameter(parts[-1], par)
                self._params[e["name"]] = (node, parts[-1], par)
                for suffix, _ in self._extra_param_specs(e):
                    mp = nn.Parameter(torch.empty(0))
                    node.register_parameter(parts[-1] + suffix, mp)
                    self._params[e["name"] + suffix] = (node, parts[-1] + suffix, mp)

        # ACORT layer sharing: a shared position is the SAME module object as the position it shares (as in the reference's
        # ModuleList of repeated modules): state_dict() lists every position, parameters() each tensor once
        for stack, share in (("encoder", self._ccfg.share_enc), ("decoder", self._ccfg.share_dec)):
            layers = self._modules["core" if self.NO_BOX else "model"]._modules[stack]._modules["layers"]
            for l in range(self.num_layers):
                if share[l] > 0:
                    layers.add_module(str(l), layers._modules[str(share[l] - 1)])

    def _arenas(self):
        return {"": "_flat"}

    def _bind(self):
        """(Re)point every Parameter / buffer at its slice of the flat arenas."""
        for e in self._entries:
            node, leaf, par = self._params[e["name"]]
            view = self._flat[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])
            if par is None:
                node._buffers[leaf] = view
            else:
                par.data = view
                for suffix, attr in self._extra_param_specs(e):
                    arena = getattr(self, attr, None)
                    if arena is None:
                        continue
                    self._params[e["name"] + suffix][2].data = arena[e["offset"]:e["offset"] + e["numel"]].view(e["shape"])

    def _apply(self, fn, recurse=True):
        for attr in self._arenas().values():
            if getattr(self, attr, None) is None:
                continue
            t = fn(getattr(self, attr))
            if t.dtype != torch.float32:
                raise TypeError("the ORT arena is fp32; choose bf16 MFMA with precision='bf16', not .half()/.bfloat16()")
            setattr(self, attr, t)
        self._bind()
        self._ws_cache = {}
        self._vr_stage = {}
        self._plans = None                 # sparse plans live on the old device
        self._ccfg.sparse_fwd = None
        self._ccfg.sparse_bwd = None
        for par in self.parameters():
            if par.grad is not None:
                par.grad = fn(par.grad)
        return self

    def named_weight_entries(self):
        return [e for e in self._entries if e["kind"] != 2]

    @torch.no_grad()
    def reset_parameters(self):
        """Same distributions as the reference: xavier-uniform on every >=2-D tensor under ``model.``
        (relation_transformer.py:336-338), torch defaults elsewhere (nn.Linear for att_embed and all biases,
        ones/zeros for LayerNorm), ``pe`` built in fp32 like transformer.py:369-374."""
        for e in self._entries:
            node, leaf, par = self._params[e["name"]]
            name, shape = e["name"], e["shape"]
            t = node._buffers[leaf] if par is None else par.data
            if e["kind"] == 2:
                d = shape[-1]
                position = torch.arange(0, shape[1]).unsqueeze(1).float()
                div_term = torch.exp(torch.arange(0, d, 2).float() * -(math.log(10000.0) / d))
                t[0, :, 0::2] = torch.sin(position * div_term)
                t[0, :, 1::2] = torch.cos(position * div_term)
            elif name.endswith(".a_2"):
                t.fill_(1.0)
            elif name.endswith(".b_2"):
                t.zero_()
            elif len(shape) >= 2:
                if name.startswith(("model.", "core.")):     # the plain transformer initialises everything under `core.` (transformer.py:660-664)
                    nn.init.xavier_uniform_(t)
                else:
                    nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            else:  # Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                wshape = self._shape_of(name[:-len("bias")] + "weight")
                bound = 1.0 / math.sqrt(wshape[1]) if wshape is not None and wshape[1] > 0 else 0.0
                t.uniform_(-bound, bound)

    def _shape_of(self, name):
        for e in self._entries:
            if e["name"] == name:
                return e["shape"]
        return None

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts the reference's checkpoint formats as they are on disk: dense, COO-sparse (``model_best_pruned_sparse.pth``,
        prune.py:200-221) and fp16-cast (``scripts/eval_model.py:73-77``) — entries are densified and cast to the fp32 arena."""
        from ..utils.model_utils import densify_state_dict
        sd = {k: (v.float() if isinstance(v, torch.Tensor) and v.is_floating_point() and v.dtype != torch.float32 else v)
              for k, v in densify_state_dict(state_dict).items()}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._plans = None          # sparse plans: block selection and capacities were taken from the OLD weights
        return out

    # ------------------------------------------------------------------ helpers
    def _eff_params_ptr(self, train, seed):
        """Device pointer of the arena the kernels read (the `_prune` variant materialises s*W first)."""
        return L.ptr(self._flat)

    def _eff_params_tensor(self):
        """The arena tensor behind the last ``_eff_params_ptr`` call."""
        return self._flat

    def enable_sparse_kernels(self, min_sparsity=0.9, train=False, fmt=None):
        """Run every weight block whose fraction of zeros is >= ``min_sparsity`` as a sparse product (``ortk_spmm``)
        instead of a dense GEMM on the zero-filled weight; ``None`` goes back to dense GEMMs.  Same results as the reference's
        dense-on-zero-filled-weights flow (scripts/eval_model.py:64-88, pruning/masked_layer.py:134-135) up to fp32 summation
        order.  Decoding (``mode="sample"``) always uses the plan; ``train=True`` also routes the teacher-forced forward and —
        in mixed precision — the data gradients of the backward through it (the weight gradients stay dense: the
        straight-through mask gradient needs them at every position).  The sparse images are rebuilt on the device inside
        every call from that call's effective weights; only the block selection and the buffer capacities are fixed here
        (from the CURRENT eval-mode weights) — see :meth:`check_sparse_overflow`.

        ``min_sparsity="auto"``: per block, the measured crossover of its shape (``sparse.CROSSOVER``, from
        ``scratch/spmm_crossover.py`` on MI355X): a block goes sparse only where the sparse product beat the dense MFMA GEMM on
        zero-filled weights — at 95 % zeros no block does, at the reference's published 98.8 % models most do."""
        self._sparse_min = min_sparsity
        self._sparse_fmt = {None: None, "ell16": L.SP_ELL16, "gu16": L.SP_GU16, "ell32": L.SP_ELL32}[fmt]      # None: by precision
        self._sparse_train = bool(train) and min_sparsity is not None
        self._plans = None
        self._ccfg.sparse_fwd = None
        self._ccfg.sparse_bwd = None
        if min_sparsity is not None and self._flat.is_cuda:
            self._sparse_plans()

    def enable_sparse_stream(self, on=True):
        """Decode pruned weights through the decoder stack kernel's SPARSE weight stream (``ORTK_DEC_SPARSE_STREAM``,
        ``include/ortk.h``): the kernel pulls only the non-zeros of the decoder weights (rebuilt on the device inside every
        ``ortk_decode`` from that call's weights) instead of the zero-filled dense matrices the reference multiplies by
        (scripts/eval_model.py:64-88).  Correct at any density; pays above ~80 % zeros.  ``on="auto"`` measures the
        decoder's zero fraction once (host sync) and switches the stream on when it is >= 0.8 — and from 98.5 % zeros on (measured crossover, scratch/gather_crossover.py) in its
        gather form (``ORTK_DEC_SPARSE_GATHER``: per-column gather lists, work proportional to the non-zeros; ``on="gather"``
        forces that form)."""
        gather = on == "gather"
        if on == "auto":
            self._eff_params_ptr(False, 0)          # (the `_prune` variant materialises s*W)
            eff = self._eff_params_tensor()
            dec = [e for e in self._entries if ".decoder.layers." in e["name"] and len(e["shape"]) >= 2]
            nz = sum(int(torch.count_nonzero(eff[e["offset"]:e["offset"] + e["numel"]])) for e in dec)
            tot = sum(e["numel"] for e in dec)
            on, gather = nz <= 0.2 * tot, nz <= 0.015 * tot
        self._sparse_stream = bool(on)
        self._sparse_gather = bool(on) and gather
        return self._sparse_stream

    def _sparse_plans(self):
        """(forward plan, backward plan) or (None, None); created on first use after enable / a device move."""
        if getattr(self, "_sparse_min", None) is None:
            return None, None
        if getattr(self, "_plans", None) is None:
            from ..sparse import make_plans
            L.require_gpu()
            self._eff_params_ptr(False, 0)
            eff = self._eff_params_tensor()
            pf, pb = make_plans(self._ccfg, eff, self._sparse_min, self.precision, backward=self._sparse_train,
                                fmt=getattr(self, "_sparse_fmt", None), density_of=self._train_density if self._sparse_train else None)
            if pf is not None:       # validate the capacities against the weights they were planned from
                pf.build(eff[:self._n_train].bfloat16() if self.precision else eff)
                pf.check_overflow()
            self._plans = (pf, pb)
            if self._sparse_train:
                self._ccfg.sparse_fwd = C.cast(pf.ref(), C.c_void_p) if pf is not None else None
                self._ccfg.sparse_bwd = C.cast(pb.ref(), C.c_void_p) if pb is not None else None
        return self._plans

    def _train_density(self, offset, N, K):
        """Expected density of a weight block under the TRAINING-mode mask sample (dense class: no masks, 0)."""
        return 0.0

    def check_sparse_overflow(self):
        """Host sync: raises if a sparse image built since the last check had to drop entries."""
        for pl in self._sparse_plans():
            if pl is not None:
                pl.check_overflow()

    def _next_seed(self):
        self._seed_counter += 1
        return (torch.initial_seed() * 1000003 + self._seed_counter) & 0xFFFFFFFFFFFFFFFF or 1

    def _workspace(self, key, nbytes, cache):
        # ONE cached buffer per kind of call ("train", "decode" + chunk of a multi-stream decode, "step"), grown to the largest
        # request seen: the geometry in `key` varies from batch to batch in a real loop (the collate pads to the batch's longest
        # region list, 10-100; the last batch of an epoch is short) and a buffer per distinct geometry — 5-7 GB each at 256 images
        # — would pile up without bound.  Every executor call carves what it needs from the front of the buffer it is handed.
        if not cache:
            return torch.empty(int(nbytes), dtype=torch.uint8, device=self._flat.device)
        slot = (key[0], key[5] if key[0] == "decode" and len(key) > 5 else 0)
        ws = self._ws_cache.get(slot)
        if ws is None or ws.numel() < nbytes:
            ws = None
            self._ws_cache.pop(slot, None)          # (release the smaller buffer before asking for the larger one)
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self._flat.device)
            self._ws_cache[slot] = ws
        return ws

    @staticmethod
    def clip_att(att_feats, att_masks, boxes, max_len=None):
        """relation_transformer.py:398-405 (boxes are cut to the same length; collate pads all three alike).  The longest
        region list of the batch is a device-to-host read of the mask (in the reference too: `.max()`), i.e. a stream
        synchronisation per call; a caller that knows it — the collate function builds the mask from the list lengths and
        returns it as ``att_max_len`` — passes it as `max_len` and nothing is read back."""
        if att_masks is not None:
            if max_len is None:
                max_len = int(att_masks.long().sum(1).max())
            if max_len < att_masks.size(1):
                att_feats, att_masks = att_feats[:, :max_len], att_masks[:, :max_len]
                boxes = boxes[:, :max_len] if boxes is not None else boxes
        return att_feats, att_masks, boxes

    def _prepare(self, att_feats, boxes, att_masks, att_max_len=None):
        L.require_gpu()
        assert att_feats.is_cuda and boxes.is_cuda, "inputs must live on the MI355X (no CPU path)"
        att_feats, att_masks, boxes = self.clip_att(att_feats, att_masks, boxes, att_max_len)
        if att_masks is None:
            att_masks = att_feats.new_ones(att_feats.shape[:2])
        assert att_feats.size(-1) == self.att_feat_size and boxes.size(-1) == 4
        return L.f32c(att_feats), L.f32c(boxes), L.f32c(att_masks)

    def _valid_rows(self, cap_len, R, T, dev):
        """Device tables of the valid-position decoder layout (``ortk_batch.cap_off / row_pos``) from the HOST-side caption
        lengths: ``cap_len[r]`` = decoder positions of caption r that carry a target (1 + index of its last non-zero target
        weight: ``len(tokens) + 1`` for BOS, tokens, EOS).  The lengths (and the row count, which sizes every launch) are the
        host's (the collate function knows them): no device read-back; one small asynchronous upload, the tables by device kernels."""
        n = torch.as_tensor(cap_len, dtype=torch.int64, device="cpu").clamp(1, T).clone()
        assert n.numel() == R, "cap_len needs one entry per caption row"
        # Row counts that are not a multiple of the GEMM tiles (256 rows) send the weight-gradient products (their reduction
        # runs over the rows) to the bounds-checked kernels: top the count up with padded positions of captions that have
        # some — computed like the reference computes them, weight zero — until it is a multiple of 256.
        extra = int((-int(n.sum())) % 256)
        if extra:
            room = (T - n).clamp(min=0)
            take = torch.minimum(room, (extra - (torch.cumsum(room, 0) - room)).clamp(min=0))
            if int(take.sum()) < extra:
                return None          # (nearly no padding in this batch: nothing to gain)
            n += take
        Mc = int(n.sum())
        # The tables are built ON THE DEVICE from the lengths: only `n` (8 bytes per caption) crosses PCIe, through a persistent ring of
        # four pinned staging buffers per (R, T) and an asynchronous copy.  Measured (scratch/valid_rows_upload.py, XE step with the
        # tables rebuilt every step, host synchronised per step): uploading the finished tables (65 KB of row indices; pinned +
        # cudaMemcpyAsync, or a blocking copy from pageable memory) makes every second or third step stall for 60-90 ms somewhere
        # later in the step (the host blocks inside HIP with the GPU idle); 10 KB of lengths + two device launches
        # (ortk_valid_position_tables): 11.6-11.8 ms per step, every step (11.5 with cached tables).  A training loop has a new batch every step, an SCST update new lengths every step.
        ring = self._vr_stage.setdefault((R, T), {"i": 0, "slots": []})
        if len(ring["slots"]) < 4:
            ring["slots"].append([torch.empty(R, dtype=torch.int64).pin_memory(), None])
            slot = ring["slots"][-1]
        else:
            slot = ring["slots"][ring["i"] % 4]
            slot[1].synchronize()             # (the upload that last used this slot: four batches ago)
        ring["i"] += 1
        slot[0].copy_(n)
        nd = slot[0].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event(); slot[1].record()
        off = torch.empty(R + 1, dtype=torch.int32, device=dev)
        rows = torch.empty(Mc, dtype=torch.int32, device=dev)
        L.check(L.lib().ortk_valid_position_tables(L.ptr(nd), R, T, L.ptr(off), L.ptr(rows), L.stream_ptr()), "ortk_valid_position_tables")
        return off, rows, Mc

    def valid_position_tables(self, data):
        """The device tables of the valid-position decoder layout for a batch dict that carries ``cap_len`` (see
        :meth:`_valid_rows`), built ONCE per batch and remembered in the dict (``data["_valid_rows"]``): call it where the
        batch is moved to the GPU (NativeTrainer does on first use).  None when the layout does not apply."""
        if "cap_len" not in data or data["cap_len"] is None:
            return None
        if "_valid_rows" not in data:
            seqs = data["seqs"]
            R, T = seqs.size(0), seqs.size(1) - 1
            B, S = data["att_feats"].shape[:2]
            ok = L.lib().ortk_valid_positions_ok(C.byref(self._ccfg), B, S, R, T)
            data["_valid_rows"] = self._valid_rows(data["cap_len"], R, T, seqs.device) if ok else None
        return data["_valid_rows"]

    def _make_batch(self, att_feats, boxes, att_masks, seqs=None, tok_weight=None, valid_rows=None, rollouts=False):
        b = L.Batch()
        b.no_pad_keys = 1 if rollouts else 0      # (sampled captions: causal mask only, as the cached passes that drew them; ortk.h)
        keep = [att_feats, boxes, att_masks]
        b.att_feats, b.boxes, b.att_masks = att_feats.data_ptr(), boxes.data_ptr(), att_masks.data_ptr()
        b.B, b.S = att_feats.shape[0], att_feats.shape[1]
        if seqs is not None:
            assert seqs.dtype == torch.long and seqs.is_cuda
            seqs = seqs.contiguous()
            assert seqs.size(0) % b.B == 0, "caption rows must be a multiple of the image count"
            b.seqs, b.seq_stride = seqs.data_ptr(), seqs.size(1)
            b.R, b.T = seqs.size(0), seqs.size(1) - 1
            keep.append(seqs)
            if tok_weight is not None:
                tok_weight = L.f32c(tok_weight)
                assert tok_weight.shape == (b.R, b.T)
                b.tok_weight = tok_weight.data_ptr()
                keep.append(tok_weight)
            if valid_rows is not None:
                # valid-position decoder (mixed precision, fused criterion): skip the padded caption positions
                off, rows, Mc = valid_rows
                assert off.numel() == b.R + 1
                b.cap_off, b.row_pos, b.Mc = off.data_ptr(), rows.data_ptr(), Mc
                keep += [off, rows]
        b._keep = keep
        return b

    # ------------------------------------------------------------------ teacher forcing
    def _run_forward(self, batch, train, seed, want_logp, cache_ws):
        lib = L.lib()
        self._sparse_plans()       # (re)attach the training plans to the config after enable / a device move
        nbytes = lib.ortk_train_workspace_bytes(C.byref(self._ccfg), batch.B, batch.S, batch.R, batch.T)
        ws = self._workspace(("train", batch.B, batch.S, batch.R, batch.T), nbytes, cache_ws)
        logp, ldv = None, 0
        if want_logp:
            ldv = (self.vocab_size + 127) // 128 * 128      # padded vocabulary: full GEMM tiles (pad log-probs unused)
            logp = torch.empty(batch.R, batch.T, ldv, device=self._flat.device)
        L.check(lib.ortk_forward(C.byref(self._ccfg), self._eff_params_ptr(train, seed), C.byref(batch), L.ptr(ws),
                                 ws.numel(), L.ptr(logp), ldv, int(train), seed, L.stream_ptr()), "ortk_forward")
        return logp, ws

    def _grad_views(self, gflat):
        out = []
        for e in self.named_weight_entries():
            out.append(gflat[e["offset"]:e["offset"] + e["numel"]].view(e["shape"]))
        return out

    def _param_list(self):
        return [self._params[e["name"]][2] for e in self.named_weight_entries()]

    def _run_backward_external(self, batch, ws, logp, dlogp, train, seed):
        lib = L.lib()
        L.check(lib.ortk_loss_external(C.byref(self._ccfg), C.byref(batch), L.ptr(ws), ws.numel(), L.ptr(logp),
                                       L.ptr(dlogp), logp.size(-1), L.stream_ptr()), "ortk_loss_external")
        gflat = torch.zeros(self._n_train, device=self._flat.device)
        L.check(lib.ortk_backward(C.byref(self._ccfg), self._eff_params_ptr(train, seed), L.ptr(gflat), C.byref(batch),
                                  L.ptr(ws), ws.numel(), int(train), seed, L.stream_ptr()), "ortk_backward")
        return self._finish_grads(gflat, train, seed)

    def _finish_grads(self, gflat, train, seed):
        return self._grad_views(gflat)

    def _forward(self, att_feats, boxes, seqs, att_masks=None, **kwargs):
        """``_forward`` (relation_transformer.py:368-372): log-probs (R, T, V), T = seqs.size(1) - 1."""
        att_feats, boxes, att_masks = self._prepare(att_feats, boxes, att_masks, kwargs.get("att_max_len"))
        # (rollouts=True — `_sample` under autograd: the rows are sampled captions, causal mask only; see ortk_batch.no_pad_keys)
        batch = self._make_batch(att_feats, boxes, att_masks, seqs, rollouts=bool(kwargs.get("rollouts", False)))
        train = bool(self.training)
        seed = self._next_seed() if train else 0
        params = self._param_list()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            logp = _ForwardFn.apply(self, batch, train, seed, *params)
        else:
            logp, _ = self._run_forward(batch, train, seed, want_logp=True, cache_ws=True)
        return logp[..., :self.vocab_size]

    # ------------------------------------------------------------------ decoding
    @staticmethod
    def _parse_length_penalty(s):
        if not s:
            return 0, 0.0
        kind, alpha = s.split("_")
        return {"wu": 1, "avg": 2}[kind], float(alpha)

    def _decode_opts(self, opt):
        """``opt`` dict of ``mode="sample"`` (transformer.py:471-561) -> (ortk_decode_opts without the sparse plan / memory, rows
        per image K, executor name)."""
        o = L.DecodeOpts()
        o.num_random_sample = int(opt.get("num_random_sample", 0))
        o.beam_size = int(opt.get("beam_size", 1))
        o.temperature = float(opt.get("temperature", 1.0))
        o.decoding_constraint = int(opt.get("decoding_constraint", 0))
        o.length_penalty, o.length_alpha = self._parse_length_penalty(opt.get("length_penalty", ""))
        o.seed = int(opt.get("seed", self._next_seed())) & 0xFFFFFFFF
        for k in ("group_size",):
            if int(opt.get(k, 1)) != 1:
                raise NotImplementedError("diverse beam groups (group_size > 1): the reference's own path raises AttributeError "
                                          "(caption_model.py:50 calls an undefined self.repeat_tensor); nothing to match")
        o.with_greedy = 1 if (o.num_random_sample > 0 and opt.get("with_greedy", False)) else 0
        # train-mode sampling (ortk_decode_opts.train): dropout on while the captions are drawn, keyed like the teacher-forced
        # pass of seed opt["drop_seed"]; default: follows model.training only when asked (opt["train_mode"])
        if opt.get("train_mode", False):
            # (with_greedy beside train-mode rows: the column-split stack kernel only — an unserved combination raises below)
            assert o.num_random_sample > 0, "train-mode sampling: multinomial rollouts"
            o.train, o.drop_seed = 1, int(opt["drop_seed"]) & 0xFFFFFFFFFFFFFFFF
        if o.num_random_sample > 0:
            assert o.beam_size < 1, f"Beam size must be < 1, saw {o.beam_size}"      # transformer.py:509
            K = o.num_random_sample + o.with_greedy
        else:
            assert o.beam_size >= 1, f"Beam size must be >= 1, saw {o.beam_size}"    # transformer.py:514
            assert o.beam_size <= self.vocab_size                                    # transformer.py:482
            K = o.beam_size
        # executor choice (ortk_decode_opts.exec_flags): opt["executor"] = "auto" | "unfused" | "stack" | "sparse_stream";
        # ORTK_DEC_STACK=0 / 2 in the environment (read here, on the host side, per call) = "unfused" / "stack"
        ex = opt.get("executor", {"0": "unfused", "2": "stack", "3": "stack_split"}.get(os.environ.get("ORTK_DEC_STACK", ""), "auto"))
        if ex == "auto" and getattr(self, "_sparse_stream", False):
            ex = "sparse_gather" if getattr(self, "_sparse_gather", False) else "sparse_stream"
        # "auto" lets decodes of <= 2 048 rows take the column-split stack kernel (fastest there) UNLESS another decode may run on
        # this GPU at the same time: its workgroups spin on each other and must all be resident (ortk.h: ORTK_DEC_SPLIT_SMALL).
        # `model.exclusive_gpu = False` (two processes on one device) or decode_streams > 1 switch that off.
        small = L.DEC_SPLIT_SMALL if (getattr(self, "exclusive_gpu", True) and int(opt.get("decode_streams", 0) or 1) <= 1) else 0
        o.exec_flags = {"auto": small, "unfused": L.DEC_UNFUSED, "stack": L.DEC_STACK, "sparse_stream": L.DEC_SPARSE_STREAM,
                        "stack_rb20": L.DEC_STACK | L.DEC_STACK_RB20, "stack_split": L.DEC_STACK | L.DEC_STACK_SPLIT,
                        "sparse_stream_rb20": L.DEC_SPARSE_STREAM | L.DEC_STACK_RB20,
                        "sparse_gather": L.DEC_SPARSE_STREAM | L.DEC_SPARSE_GATHER}[ex] | (int(opt.get("stack_debug", 0)) & 0xFF) << 8
        return o, K, ex

    def decode_supported(self, B, S, opt, att_max_len=None):
        """Whether ``mode="sample"`` serves this option combination for B images of S regions (e.g. train-mode rollouts with the
        greedy baseline as eval-mode rows of the same launches: the column-split stack kernel only).  No device work."""
        o, _, ex = self._decode_opts(dict(opt, seed=opt.get("seed", 0)))      # (a probe draws no seed)
        if getattr(self, "_plans", None) is not None and self._plans[0] is not None and not ex.startswith("sparse_") and not o.train:
            o.sparse = self._plans[0].ref()
        if att_max_len is not None:
            S = min(int(S), int(att_max_len))
        return L.lib().ortk_decode_workspace_bytes(C.byref(self._ccfg), int(B), int(S), C.byref(o)) != 0

    @torch.no_grad()
    def _decode(self, att_feats, boxes, att_masks, opt):
        lib = L.lib()
        o, K, ex = self._decode_opts(opt)
        B, S = att_feats.shape[:2]
        dev = self._flat.device
        seq = torch.empty(B, K, self.seq_length, dtype=torch.long, device=dev)
        lp = torch.empty(B, K, self.seq_length, device=dev)
        score = torch.empty(B, K, device=dev)
        pptr = self._eff_params_ptr(False, 0)
        fresh_plan = getattr(self, "_plans", None) is None
        plan = self._sparse_plans()[0]
        # (train-mode rollouts — the default SCST step of a pruned model with enable_sparse_kernels() — have no sparse form: every
        #  dropout site is an epilogue of the dense kernels; they run the dense products on the zero-filled effective weights, which
        #  is what the reference's MaskedLinear computes, pruning/masked_layer.py:134-135)
        if plan is not None and not ex.startswith("sparse_") and not o.train:
            o.sparse = plan.ref()
        else:
            plan = None
        beam = o.beam_size > 1 and o.num_random_sample <= 0
        # Images are independent: `opt["decode_streams"] = n` decodes the batch as n chunks on n streams, each driven by its
        # own host thread (ctypes releases the GIL) — same tokens as one call (the Gumbel hash takes the global row).
        # MEASURED on the 1 024-image beam-5 decode: 33.8 ms with 1 stream, 33.9 with 2, 51 with 3 (the HIP runtime
        # serialises the launching threads); with the decoder stack kernel (round 2) 20.6 ms with 1 stream, 20.8 with 2, 28.1 with
        # 3 (`bench.py --workload decode --decode-streams n`) — so the default stays 1; the option remains for hosts that want to pipeline.
        n = int(opt.get("decode_streams", 0)) or 1
        n = max(1, min(n, B))
        if plan is not None and n > 1:
            n = 1       # every chunk's ortk_decode rebuilds the plan's shared buffers on its own stream: not concurrently
        # opt["memory"]: device address of the encoder memory of THESE images ((B*S, d_model) rows in the activation type of the
        # precision), e.g. the training forward's (NativeTrainer.scst_step): the decode skips its own encoder pass
        if opt.get("memory"):
            o.memory = int(opt["memory"])
            n = 1

        used_ws = []

        def run(i, b0, b1, stream_ptr, out):
            oi = L.DecodeOpts.from_buffer_copy(o)
            oi.sample_row_offset = int(opt.get("sample_row_offset", 0)) + b0 * K
            nb = lib.ortk_decode_workspace_bytes(C.byref(self._ccfg), b1 - b0, S, C.byref(oi))
            if nb == 0:
                out[i] = -1
                return
            ws = self._workspace(("decode", b1 - b0, S, K, beam, i), nb, True)
            used_ws.append(ws)
            out[i] = lib.ortk_decode(C.byref(self._ccfg), pptr, L.ptr(att_feats[b0:b1]), L.ptr(boxes[b0:b1]),
                                     L.ptr(att_masks[b0:b1]), b1 - b0, S, C.byref(oi), L.ptr(ws), ws.numel(), L.ptr(seq[b0:b1]),
                                     L.ptr(lp[b0:b1]), L.ptr(score[b0:b1]), stream_ptr)

        rc = [0] * n
        if n == 1:
            run(0, 0, B, L.stream_ptr(), rc)
        else:
            import threading
            cur = torch.cuda.current_stream()
            if len(getattr(self, "_dec_streams", [])) < n:
                self._dec_streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
            streams = self._dec_streams[:n]
            bounds = [B * i // n for i in range(n + 1)]
            for s_ in streams:
                s_.wait_stream(cur)
            ths = [threading.Thread(target=run, args=(i, bounds[i], bounds[i + 1], C.c_void_p(streams[i].cuda_stream), rc))
                   for i in range(n)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            for s_ in streams:
                cur.wait_stream(s_)
        for r in rc:
            if r == -1 and n > 1:
                raise L.OrtkError("unsupported decode options")
            L.check(r, "ortk_decode")
        if plan is not None and fresh_plan:
            plan.check_overflow()      # (host sync, first decode on a new plan: the images were built from THIS call's weights)
        # The column-split stack kernel's exchanges are bounded waits (ortk.h: ORTK_DEC_SPLIT_SMALL): a decode whose groups never
        # met (another kernel held compute units) has poisoned outputs and a status word.  Reading it is a host synchronisation, so
        # it is read on the first decodes of a model, then every 64th, and whenever opt["check_status"] asks.
        if (o.exec_flags & (L.DEC_SPLIT_SMALL | L.DEC_STACK_SPLIT)) and ex != "unfused":
            self._decode_calls = getattr(self, "_decode_calls", 0) + 1
            self._status_ws = (getattr(self, "_status_ws", []) + used_ws)[-4:]       # unchecked decodes (the most recent few)
            if opt.get("check_status", self._decode_calls <= 2 or self._decode_calls % 64 == 0):
                self.check_decode_status()
        return seq, lp, score

    def check_decode_status(self):
        """Raise if one of the not yet checked decodes of this model (the most recent four at most) ran the column-split stack kernel
        and one of its exchange groups never met (its outputs are then poisoned: all-pad captions, NaN log-probs).  A host synchronisation — free for a caller that has already
        waited for the decode's output (``NativeTrainer.scst_step`` with a host-side reward)."""
        pending, self._status_ws = getattr(self, "_status_ws", []), []
        for ws in pending:
            L.check(L.lib().ortk_decode_status(L.ptr(ws), L.stream_ptr()), "ortk_decode (status)")

    def _sample(self, att_feats, boxes, att_masks=None, opt=None, **kwargs):
        """``_sample`` (relation_transformer.py:390-396) + ``_generate_captions`` (transformer.py:471-561).

        Returns ``seq (N,K,L)`` int64 and ``seq_logprobs (N,K,L)``.  Under autograd with
        ``num_random_sample > 0`` (the SCST rollout, utils/training.py:224-237) the tokens are drawn without a
        graph and their log-probs are recomputed by ONE differentiable teacher-forced pass — identical values
        (SURVEY.md §9.3), far cheaper backward than the reference's 18-step incremental graph.
        """
        opt = {} if opt is None else opt
        feats, bxs, masks = self._prepare(att_feats, boxes, att_masks, kwargs.get("att_max_len"))
        seq, lp, score = self._decode(feats, bxs, masks, opt)
        self._last_decode = (seq, lp, score, int(opt.get("beam_size", 1)))
        self.done_beams = None
        ns = int(opt.get("num_random_sample", 0))
        params = self._param_list()
        if ns > 0 and torch.is_grad_enabled() and any(p.requires_grad for p in params):
            rows = seq.view(-1, self.seq_length)
            tf_in = torch.cat([rows.new_full((rows.size(0), 1), self.bos_idx), rows], 1)
            logp = self._forward(feats, bxs, tf_in, masks, rollouts=True)      # (N*ns, L, V), differentiable
            tok_lp = logp.gather(2, rows.unsqueeze(2)).squeeze(2).view_as(lp)
            lp = torch.where(seq != self.pad_idx, tok_lp, lp)
        return seq, lp

    @property
    def beams(self):
        """Lazy equivalent of the reference's ``done_beams`` (caption_model.py:221-226): per image a list of
        ``{"seq", "logps" (per-token), "p"}`` — built on first access (host sync)."""
        if self.done_beams is None and getattr(self, "_last_decode", None) is not None:
            seq, lp, score, beam = self._last_decode
            out = []
            for n in range(seq.size(0)):
                cur = []
                for k in range(seq.size(1)):
                    ln = int((seq[n, k] != 0).sum())
                    cur.append({"seq": seq[n, k, :ln], "logps": lp[n, k, :ln], "p": float(score[n, k])})
                out.append(cur)
            self.done_beams = out
        return self.done_beams

    @torch.no_grad()
    def get_logprobs_state(self, it, memory, mask, state):
        """One cached-attention decoder step — ``get_logprobs_state`` (relation_transformer.py:374-387).

        ``it`` (rows,) int64, ``memory`` (rows, S, d) from :meth:`encode`, ``mask`` (rows, S) or (rows, 1, S), ``state`` =
        ``None`` on the first call, afterwards the list returned by the previous call (possibly re-ordered along dim 1 by a
        beam search, caption_model.py:106-110).  Returns ``(logp (rows, V), state)`` with the reference's state layout
        (transformer.py:457-469): ``[ys (1, rows, 1)]`` + per decoder layer ``self K, self V (h, rows, t+1, d_k)``,
        ``src K, src V (h, rows, S, d_k)``.  The position is the length of the self-attention cache (the reference keeps it
        in a module counter reset by ``reset_cache``).  This is the API-compatible path: every call re-packs the caches;
        ``mode="sample"`` runs the whole loop on the device."""
        lib = L.lib()
        rows, S, d = memory.shape
        Lr, H = self.num_layers, self.num_heads
        dk, T = d // H, self.seq_length
        dev = self._flat.device
        # layout of the projected memory (ortk_project_memory): one cw-wide slice per DISTINCT decoder layer (ACORT layer
        # sharing, ortk_config.share_dec), cw = d when the module shares its key / value projection ("kv": K = V) else [K | V]
        share = [int(self._ccfg.share_dec[l]) for l in range(Lr)]
        distinct = [l for l in range(Lr) if share[l] == 0]
        slot = [distinct.index(l if share[l] == 0 else share[l] - 1) for l in range(Lr)]
        U = len(distinct)
        kv_shared = int(self._ccfg.share_att_dec) == 1
        cw, vo = (d, 0) if kv_shared else (2 * d, d)
        it = it.to(dev).long().contiguous()
        masks = mask.reshape(rows, S).to(dev).float().contiguous()
        pptr = self._eff_params_ptr(False, 0)
        nbytes = lib.ortk_decode_step_workspace_bytes(C.byref(self._ccfg), rows)
        ws = self._workspace(("step", rows), nbytes, True)
        self_k = torch.zeros(Lr, rows, T, d, device=dev)
        self_v = torch.zeros(Lr, rows, T, d, device=dev)
        if state is None:
            t = 0
            ckv = torch.empty(rows * S, U * cw, device=dev)
            mem = memory.to(dev).float().contiguous()
            L.check(lib.ortk_project_memory(C.byref(self._ccfg), pptr, L.ptr(mem), rows * S, L.ptr(ws), ws.numel(),
                                            L.ptr(ckv), L.stream_ptr()), "ortk_project_memory")
        else:
            caches = state[1:]
            assert len(caches) == 4 * Lr, "state must come from get_logprobs_state"
            t = caches[0].size(2)
            assert t < T, "cache is full"
            unhead = lambda c: c.permute(1, 2, 0, 3).reshape(rows, c.size(2), d)      # (h, rows, len, dk) -> (rows, len, d)
            ckv = torch.empty(rows, S, U, cw, device=dev)
            for l in range(Lr):
                self_k[l, :, :t] = unhead(caches[4 * l])
                self_v[l, :, :t] = unhead(caches[4 * l + 1])
                ckv[:, :, slot[l], 0:d] = unhead(caches[4 * l + 2])
                if not kv_shared:
                    ckv[:, :, slot[l], d:2 * d] = unhead(caches[4 * l + 3])
            ckv = ckv.view(rows * S, U * cw)
        logp = torch.empty(rows, self.vocab_size, device=dev)
        L.check(lib.ortk_decode_step(C.byref(self._ccfg), pptr, L.ptr(it), t, rows, rows, S, L.ptr(ckv), L.ptr(masks),
                                     L.ptr(self_k), L.ptr(self_v), T, L.ptr(ws), ws.numel(), L.ptr(logp), self.vocab_size,
                                     L.stream_ptr()), "ortk_decode_step")
        head = lambda x: x.reshape(rows, -1, H, dk).permute(2, 0, 1, 3).contiguous()  # (rows, len, d) -> (h, rows, len, dk)
        ckv4 = ckv.view(rows, S, U, cw)
        new_state = [it.view(1, rows, 1)]
        for l in range(Lr):
            new_state += [head(self_k[l, :, :t + 1]), head(self_v[l, :, :t + 1]), head(ckv4[:, :, slot[l], 0:d]),
                          head(ckv4[:, :, slot[l], vo:vo + d])]
        return logp, new_state

    @torch.no_grad()
    def encode(self, att_feats, boxes, att_masks=None):
        """Encoder memory (B, S, d) — ``model.encode`` (relation_transformer.py:69-70) incl. feature prep."""
        lib = L.lib()
        feats, bxs, masks = self._prepare(att_feats, boxes, att_masks)
        B, S = feats.shape[:2]
        o = L.DecodeOpts()
        o.beam_size, o.temperature = 1, 1.0
        ws = self._workspace(("decode", B, S, 1, False), lib.ortk_decode_workspace_bytes(C.byref(self._ccfg), B, S, C.byref(o)), True)
        mem = torch.empty(B, S, self.d_model, device=self._flat.device)
        L.check(lib.ortk_encode(C.byref(self._ccfg), self._eff_params_ptr(False, 0), L.ptr(feats), L.ptr(bxs), L.ptr(masks),
                                B, S, L.ptr(ws), ws.numel(), L.ptr(mem), L.stream_ptr()), "ortk_encode")
        return mem

    @staticmethod
    def add_argparse_args(parser):
        """Model flags of transformer.py:563-614 and relation_transformer.py:414-426."""
        RelationTransformerModel.COLLATE_FN.add_argparse_args(parser)      # the data flags, as the reference does (relation_transformer.py:417)
        parser.add_argument("--d_model", type=int, default=512)
        parser.add_argument("--dim_feedforward", type=int, default=2048)
        parser.add_argument("--num_layers", type=int, default=6)
        parser.add_argument("--num_heads", type=int, default=8)
        parser.add_argument("--drop_prob_src", type=float, default=0.5)
        parser.add_argument("--att_feat_size", type=int, default=2048)
        for k in ("share_att_encoder", "share_att_decoder", "share_layer_encoder", "share_layer_decoder"):
            parser.add_argument("--" + k, default=None)
        parser.add_argument("--no_box_trigonometric_embedding", action="store_true")
        parser.add_argument("--ortk_precision", type=str, default="fp32", choices=("fp32", "bf16"))
