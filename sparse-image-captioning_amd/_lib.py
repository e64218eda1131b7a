"""ctypes binding of libortk.so (the C-ABI declared in include/ortk.h).

The HIP library is the product: there is NO Python / PyTorch fallback for any op.  ``lib()`` raises
``OrtkUnavailable`` loudly when the shared object is missing, and every device entry point refuses to run
without a gfx950 GPU.  PyTorch is used only for device memory (``tensor.data_ptr()``), the current HIP stream
and ``torch.distributed``.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libortk.so")


class OrtkUnavailable(RuntimeError):
    pass


class OrtkError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("d_ff", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
                ("vocab", C.c_int32), ("feat", C.c_int32), ("seq_len", C.c_int32),
                ("pad_id", C.c_int32), ("bos_id", C.c_int32), ("eos_id", C.c_int32), ("unk_id", C.c_int32),
                ("box_trig", C.c_int32), ("precision", C.c_int32), ("drop_src", C.c_float), ("drop", C.c_float),
                ("share_enc", C.c_int32 * 16), ("share_dec", C.c_int32 * 16),
                ("share_att_enc", C.c_int32), ("share_att_dec", C.c_int32), ("no_box", C.c_int32),
                ("sparse_fwd", C.c_void_p), ("sparse_bwd", C.c_void_p)]


class Batch(C.Structure):
    _fields_ = [("att_feats", C.c_void_p), ("boxes", C.c_void_p), ("att_masks", C.c_void_p), ("seqs", C.c_void_p),
                ("seq_stride", C.c_int64), ("tok_weight", C.c_void_p),
                ("B", C.c_int32), ("S", C.c_int32), ("R", C.c_int32), ("T", C.c_int32),
                ("cap_off", C.c_void_p), ("row_pos", C.c_void_p), ("Mc", C.c_int32), ("no_pad_keys", C.c_int32)]


SP_ELL32, SP_ELL16, SP_GU16 = 0, 1, 2     # ortk_sparse_plan.format


class EllBlock(C.Structure):
    _fields_ = [("src_offset", C.c_int64), ("ld", C.c_int64), ("stream_offset", C.c_int64), ("capacity", C.c_int64),
                ("N", C.c_int32), ("K", C.c_int32), ("chunk0", C.c_int32), ("row0", C.c_int32)]


class EllPlanStruct(C.Structure):
    _fields_ = [("blocks_host", C.POINTER(EllBlock)), ("blocks_dev", C.c_void_p), ("nblocks", C.c_int32), ("format", C.c_int32),
                ("stream", C.c_void_p), ("chunk_ptr", C.c_void_p), ("chunk_len", C.c_void_p), ("perm", C.c_void_p),
                ("count_scratch", C.c_void_p), ("overflow", C.c_void_p), ("total_rows", C.c_int64)]


class SpmmArgs(C.Structure):
    _fields_ = [("X", C.c_void_p), ("Y", C.c_void_p), ("ldx", C.c_int64), ("ldy", C.c_int64), ("M", C.c_int64),
                ("x_dtype", C.c_int32), ("y_dtype", C.c_int32),
                ("bias", C.c_void_p), ("rowscale", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int64),
                ("gate", C.c_void_p), ("ldg", C.c_int64), ("gate_dtype", C.c_int32), ("gate_scale", C.c_float),
                ("relu", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32), ("drop_rows", C.c_void_p)]


class ChainUnit(C.Structure):
    _fields_ = [("offset", C.c_int64), ("ld", C.c_int64)]


class ChainArgs(C.Structure):
    _fields_ = [("w16", C.c_void_p), ("units_dev", C.c_void_p), ("n_units", C.c_int32), ("packed", C.c_void_p), ("packed_bytes", C.c_size_t),
                ("M", C.c_int64), ("x_in", C.c_void_p),
                ("a_in", C.c_void_p), ("bias_r", C.c_void_p), ("x_mid", C.c_void_p), ("seed_r", C.c_uint32),
                ("g1", C.c_void_p), ("b1", C.c_void_p), ("y1", C.c_void_p), ("st1", C.c_void_p),
                ("n1", C.c_int32), ("bias_s1", C.c_void_p), ("out1", C.c_void_p), ("ld1", C.c_int64),
                ("NC", C.c_int32), ("bias_h", C.c_void_p), ("bias_o", C.c_void_p), ("h", C.c_void_p), ("x_out", C.c_void_p),
                ("seed_h", C.c_uint32), ("seed_o", C.c_uint32),
                ("g2", C.c_void_p), ("b2", C.c_void_p), ("y2", C.c_void_p), ("st2", C.c_void_p),
                ("n2", C.c_int32), ("bias_s2", C.c_void_p), ("out2", C.c_void_p), ("ld2", C.c_int64),
                ("drop_p", C.c_float), ("eps", C.c_float), ("progress", C.c_void_p), ("drop_rows", C.c_void_p)]


class BChainArgs(C.Structure):
    _fields_ = [("w16t", C.c_void_p), ("units_dev", C.c_void_p), ("n_units", C.c_int32), ("packed", C.c_void_p), ("packed_bytes", C.c_size_t),
                ("M", C.c_int64),
                ("nin", C.c_int32), ("ain", C.c_void_p), ("ld_ain", C.c_int64), ("dz0", C.c_void_p),
                ("xa", C.c_void_p), ("sta", C.c_void_p), ("ga", C.c_void_p), ("dresa", C.c_void_p), ("dxa", C.c_void_p), ("daa", C.c_void_p),
                ("dba", C.c_void_p), ("dza", C.c_void_p), ("seed_a", C.c_uint32), ("mask_a", C.c_int32),
                ("NC", C.c_int32), ("hgate", C.c_void_p), ("gh", C.c_void_p), ("gate_scale", C.c_float),
                ("xb", C.c_void_p), ("stb", C.c_void_p), ("gb", C.c_void_p), ("dresb", C.c_void_p), ("dxb", C.c_void_p), ("dab", C.c_void_p),
                ("dbb", C.c_void_p), ("dzb", C.c_void_p), ("seed_b", C.c_uint32), ("mask_b", C.c_int32),
                ("n2", C.c_int32), ("out2", C.c_void_p), ("drop_p", C.c_float), ("eps", C.c_float), ("drop_rows", C.c_void_p)]


class MaskedAdamArgs(C.Structure):
    _fields_ = [("w", C.c_void_p), ("g", C.c_void_p), ("mw", C.c_void_p), ("vw", C.c_void_p),
                ("ml", C.c_void_p), ("mm", C.c_void_p), ("mv", C.c_void_p),
                ("draws", C.c_void_p), ("active", C.c_void_p), ("extra_coef", C.c_void_p),
                ("n", C.c_int64), ("index0", C.c_int64), ("mode", C.c_int32), ("seed", C.c_uint32),
                ("lr_w", C.c_float), ("eps_w", C.c_float), ("lr_m", C.c_float), ("eps_m", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("clip", C.c_float), ("bc1", C.c_float), ("bc2", C.c_float)]


class Tuning(C.Structure):
    _fields_ = [("gemm_impl", C.c_int32), ("gemm_t64", C.c_int32), ("attn_impl", C.c_int32), ("attn16_min_lq", C.c_int32),
                ("side_stream", C.c_int32), ("row_chain", C.c_int32), ("chain_wide", C.c_int32), ("spmm_alias", C.c_int32), ("f32_split", C.c_int32), ("wgrad_wgs", C.c_int32),
                ("wgrad_group", C.c_int32), ("wgrad_group_splitk", C.c_int32), ("wgrad_group_wgs", C.c_int32), ("wgrad_group_tail", C.c_int32), ("feats_bf16", C.c_int32), ("ln_fuse", C.c_int32), ("samp_epilogue", C.c_int32), ("gemm_epilogue", C.c_int32)]


DEC_UNFUSED, DEC_STACK, DEC_SPARSE_STREAM, DEC_STACK_RB20, DEC_STACK_SPLIT, DEC_SPLIT_SMALL, DEC_SPARSE_GATHER = 1, 2, 4, 8, 16, 32, 64      # ortk_decode_opts.exec_flags


class DecodeOpts(C.Structure):
    _fields_ = [("beam_size", C.c_int32), ("num_random_sample", C.c_int32), ("temperature", C.c_float),
                ("decoding_constraint", C.c_int32), ("length_penalty", C.c_int32), ("length_alpha", C.c_double),
                ("seed", C.c_uint64), ("sparse", C.POINTER(EllPlanStruct)), ("exec_flags", C.c_int32), ("with_greedy", C.c_int32), ("sample_row_offset", C.c_int64),
                ("train", C.c_int32), ("drop_seed", C.c_uint64), ("memory", C.c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("transA", C.c_int32), ("transB", C.c_int32),
                ("bias", C.c_void_p), ("rowscale", C.c_void_p), ("resid", C.c_void_p), ("ldr", C.c_int64),
                ("gate", C.c_void_p), ("ldg", C.c_int64), ("gate_scale", C.c_float),
                ("relu", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("accumulate", C.c_int32), ("splitk", C.c_int32), ("precision", C.c_int32),
                ("a_dtype", C.c_int32), ("b_dtype", C.c_int32), ("c_dtype", C.c_int32), ("gate_dtype", C.c_int32),
                ("colsum", C.c_void_p), ("drop_row_stride", C.c_int32), ("drop_row_off", C.c_int32),
                ("ln_mode", C.c_int32), ("ln_y_dtype", C.c_int32),
                ("ln_a", C.c_void_p), ("ln_b", C.c_void_p), ("ln_y", C.c_void_p), ("ln_stats", C.c_void_p), ("ln_eps", C.c_float),
                ("ln_x", C.c_void_p), ("ln_dres", C.c_void_p), ("ln_da", C.c_void_p), ("ln_db", C.c_void_p),
                ("tile_stats", C.c_void_p), ("stat_ncols", C.c_int32), ("drop_rows", C.c_void_p),
                ("tile_samp", C.c_void_p), ("samp_seq", C.c_void_p), ("samp_seed", C.c_uint64), ("samp_row_offset", C.c_int64),
                ("samp_L", C.c_int32), ("samp_t", C.c_int32), ("samp_greedy_stride", C.c_int32), ("samp_sample", C.c_int32),
                ("samp_fast", C.c_int32), ("samp_no_store", C.c_int32), ("samp_inv_temperature", C.c_float)]


class WgradItem(C.Structure):
    _fields_ = [("dY", C.c_void_p), ("lddy", C.c_int64), ("X", C.c_void_p), ("ldx", C.c_int64),
                ("dW", C.c_void_p), ("lddw", C.c_int64), ("db", C.c_void_p), ("Nout", C.c_int32), ("Kin", C.c_int32)]


WGRAD_MAX = 8


class WgradGroupArgs(C.Structure):
    _fields_ = [("item", WgradItem * WGRAD_MAX), ("n", C.c_int32), ("splitk", C.c_int32), ("rows", C.c_int64), ("flags", C.c_int32),
                ("ws", C.c_void_p), ("ws_bytes", C.c_size_t)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
                ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
                ("kmask", C.c_void_p), ("bias", C.c_void_p), ("kv_index", C.c_void_p), ("kv_group_stride", C.c_int64),
                ("p", C.c_void_p),
                ("nkv", C.c_int32), ("H", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32), ("dk", C.c_int32),
                ("causal_period", C.c_int32), ("drop_p", C.c_float), ("drop_seed", C.c_uint32),
                ("d_o", C.c_void_p), ("dq", C.c_void_p), ("d_k", C.c_void_p), ("dv", C.c_void_p), ("dscore", C.c_void_p),
                ("lddo", C.c_int64), ("lddq", C.c_int64), ("lddk", C.c_int64), ("lddv", C.c_int64),
                ("o_dtype", C.c_int32), ("dqkv_dtype", C.c_int32), ("kv_dtype", C.c_int32),
                ("k_new", C.c_void_p), ("v_new", C.c_void_p), ("ld_new", C.c_int64), ("bwd_part", C.c_int32),
                ("precision", C.c_int32), ("qkv_dtype", C.c_int32),
                ("q_off", C.c_void_p), ("q_off_stride", C.c_int32), ("kv_ragged", C.c_int32),
                ("drop_tf_T", C.c_int32), ("drop_tf_t", C.c_int32), ("drop_tf_lk", C.c_int32), ("drop_rows", C.c_void_p)]


_P, _I32, _I64, _F, _U32, _U64, _SZ = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_uint64, C.c_size_t
_CFG = C.POINTER(Config)

# name -> (restype, argtypes); kept in sync with include/ortk.h by tests/test_lib_host.py
SIGNATURES = {
    "ortk_version": (_I32, []),
    "ortk_device_ok": (_I32, []),
    "ortk_arena_numel": (_I64, [_CFG]),
    "ortk_arena_numel_with_buffers": (_I64, [_CFG]),
    "ortk_arena_entries": (_I32, [_CFG]),
    "ortk_arena_entry": (_I32, [_CFG, _I32, C.c_char_p, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I32),
                               C.POINTER(_I64), C.POINTER(_I32)]),
    "ortk_train_workspace_bytes": (_SZ, [_CFG, _I32, _I32, _I32, _I32]),
    "ortk_valid_positions_ok": (_I32, [_CFG, _I32, _I32, _I32, _I32]),
    "ortk_forward": (_I32, [_CFG, _P, C.POINTER(Batch), _P, _SZ, _P, _I64, _I32, _U64, _P]),
    "ortk_forward_phase": (_I32, [_CFG, _P, C.POINTER(Batch), _P, _SZ, _P, _I64, _I32, _U64, _I32, _P]),
    "ortk_train_workspace_memory": (_P, [_CFG, _I32, _I32, _I32, _I32, _P, C.POINTER(C.c_int32)]),
    "ortk_loss": (_I32, [_CFG, C.POINTER(Batch), _P, _SZ, _P, _P, _P]),
    "ortk_loss_external": (_I32, [_CFG, C.POINTER(Batch), _P, _SZ, _P, _P, _I64, _P]),
    "ortk_backward": (_I32, [_CFG, _P, _P, C.POINTER(Batch), _P, _SZ, _I32, _U64, _P]),
    "ortk_backward_phase": (_I32, [_CFG, _P, _P, C.POINTER(Batch), _P, _SZ, _I32, _U64, _I32, _P]),
    "ortk_arena_decoder_offset": (_I64, [_CFG]),
    "ortk_decode_workspace_bytes": (_SZ, [_CFG, _I32, _I32, C.POINTER(DecodeOpts)]),
    "ortk_decode": (_I32, [_CFG, _P, _P, _P, _P, _I32, _I32, C.POINTER(DecodeOpts), _P, _SZ, _P, _P, _P, _P]),
    "ortk_decode_status": (_I32, [_P, _P]),
    "ortk_encode": (_I32, [_CFG, _P, _P, _P, _P, _I32, _I32, _P, _SZ, _P, _P]),
    "ortk_chain_packed_bytes": (_SZ, [_I32]),
    "ortk_row_chain": (_I32, [C.POINTER(ChainArgs), _P]),
    "ortk_row_bchain": (_I32, [C.POINTER(BChainArgs), _P]),
    "ortk_get_tuning": (None, [C.POINTER(Tuning)]),
    "ortk_set_tuning": (_I32, [C.POINTER(Tuning)]),
    "ortk_gemm": (_I32, [C.POINTER(GemmArgs), _P]),
    "ortk_wgrad_group": (_I32, [C.POINTER(WgradGroupArgs), _P]),
    "ortk_wgrad_group_workspace_bytes": (_SZ, [C.POINTER(WgradGroupArgs)]),
    "ortk_prof_enable": (_I32, [_I32]),
    "ortk_prof_collect": (_I32, [_I32, C.POINTER(_I64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ortk_prof_collect_bytes": (_I32, [_I32, C.POINTER(C.c_double)]),
    "ortk_prof_collect_units": (_I32, [_I32, C.POINTER(C.c_double)]),
    "ortk_layernorm_fwd": (_I32, [_P, _P, _P, _P, _I32, _P, _I64, _I32, _F, _P]),
    "ortk_layernorm_bwd": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F, _P]),
    "ortk_layernorm_bwd_drop": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F, _P, _I32, _F, C.c_uint32, _P]),
    "ortk_layernorm_bwd_drop_rows": (_I32, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F, _P, _I32, _F, C.c_uint32, _P, _P]),
    "ortk_layernorm_bwd_dt": (_I32, [_P, _I32, _P, _P, _P, _P, _P, _P, _P, _I64, _I32, _F, _P, _I32, _F, C.c_uint32, _P, _P]),
    "ortk_box_logbias_fwd": (_I32, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_F), _P, _I32, _I32, _I32, _I32, _P]),
    "ortk_box_logbias_bwd": (_I32, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_F), _P, C.POINTER(_P), C.POINTER(_P),
                                   _I32, _I32, _I32, _I32, _P]),
    "ortk_box_embedding": (_I32, [_P, C.POINTER(_F), _P, _I32, _I32, _P]),
    "ortk_attention_fwd": (_I32, [C.POINTER(AttnArgs), _P]),
    "ortk_attention_bwd": (_I32, [C.POINTER(AttnArgs), _P]),
    "ortk_valid_position_tables": (_I32, [_P, _I32, _I32, _P, _P, _P]),
    "ortk_embed_fwd": (_I32, [_P, _I64, _P, _P, _P, _P, _I64, _I32, _I32, _I32, _I32, _F, _U32, _P]),
    "ortk_embed_bwd": (_I32, [_P, _I64, _P, _P, _I64, _I32, _I32, _F, _U32, _P]),
    "ortk_log_softmax": (_I32, [_P, _I64, _I32, _I64, _F, _P]),
    "ortk_xent_scratch_floats": (_I64, [_I64]),
    "ortk_xent_fwd_bwd": (_I32, [_P, _P, _I64, _I32, _P, _P, _P, _P, _I64, _I32, _I64, _P, _I32, _I64, _P]),
    "ortk_log_softmax_bwd": (_I32, [_P, _P, _I64, _P, _I32, _I64, _I64, _I32, _P]),
    "ortk_colsum": (_I32, [_P, _I32, _I64, _P, _I64, _I32, _P]),
    "ortk_gate_apply": (_I32, [_P, _P, _P, _I32, _I64, _F, _P]),
    "ortk_dropout_apply": (_I32, [_P, _P, _I32, _I64, _F, _U32, _P]),
    "ortk_dropout_apply_rows": (_I32, [_P, _P, _I32, _I64, _I32, _F, _U32, _P, _P]),
    "ortk_dropout_site_seed": (_U32, [_U64, _I32, _I32, _I32]),
    "ortk_cast_bf16": (_I32, [_P, _P, _I64, _P]),
    "ortk_fill": (_I32, [_P, _I64, _F, _P]),
    "ortk_sum_scratch_floats": (_I64, [_I64]),
    "ortk_sum": (_I32, [_P, _I64, _P, _P, _P]),
    "ortk_adam_clip": (_I32, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _P]),
    "ortk_adam_clip_zero": (_I32, [_P, _P, _P, _P, _I64, _F, _F, _F, _F, _F, _F, _F, _P]),
    "ortk_mask_apply": (_I32, [_P, _P, _P, _I64, _I32, _U32, _P]),
    "ortk_mask_bwd": (_I32, [_P, _P, _P, _P, _P, _I64, _I32, _U32, _P, _P]),
    "ortk_masked_adam_step": (_I32, [C.POINTER(MaskedAdamArgs), _P]),
    "ortk_mask_count": (_I32, [_P, _I64, _I32, _P, _P]),
    "ortk_mask_apply_draws": (_I32, [_P, _P, _P, _P, _I64, _P]),
    "ortk_mask_bwd_draws": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _P, _P]),
    "ortk_decode_step_workspace_bytes": (_SZ, [_CFG, _I32]),
    "ortk_project_memory": (_I32, [_CFG, _P, _P, _I64, _P, _SZ, _P, _P]),
    "ortk_decode_step": (_I32, [_CFG, _P, _P, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _I32, _P, _SZ, _P, _I64, _P]),
    "ortk_axpy_cols": (_I32, [_P, _P, _I32, _I64, _I64, _I32, _P]),
    "ortk_linear_block": (_I32, [_CFG, _I32, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "ortk_sparse_build": (_I32, [C.POINTER(EllPlanStruct), _P, _I32, _P]),
    "ortk_spmm": (_I32, [C.POINTER(EllPlanStruct), _I32, C.POINTER(SpmmArgs), _P]),
}

_lib = None
ABI_VERSION = 2      # include/ortk.h: ORTK_VERSION


def lib():
    """The loaded library; raises OrtkUnavailable if libortk.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OrtkUnavailable(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C sparse-image-captioning_amd/csrc`).  There is no CPU / PyTorch fallback.")
        h = C.CDLL(LIB_PATH)
        h.ortk_version.restype = C.c_int32
        if h.ortk_version() != ABI_VERSION:
            raise OrtkUnavailable(
                f"{LIB_PATH} is ABI version {h.ortk_version()}, these bindings were written for version {ABI_VERSION} (include/ortk.h: "
                "ORTK_VERSION): a stale build.  Rebuild it with `make -C sparse-image-captioning_amd/csrc`.")
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def set_tuning(**kw):
    """Measurement switches (include/ortk.h: ortk_tuning), e.g. ``set_tuning(side_stream=0)``; returns the previous values."""
    t = Tuning()
    lib().ortk_get_tuning(C.byref(t))
    old = {f: getattr(t, f) for f, _ in Tuning._fields_}
    for k, v in kw.items():
        assert k in old, k
        setattr(t, k, int(v))
    check(lib().ortk_set_tuning(C.byref(t)), "ortk_set_tuning")
    return old


def require_gpu():
    """Fail loudly unless a gfx950 device is usable."""
    if not torch.cuda.is_available():
        raise OrtkUnavailable("no HIP device visible: the ORT hot path runs only on MI355X (gfx950); no CPU fallback")
    if not lib().ortk_device_ok():
        raise OrtkUnavailable("the visible device is not gfx950; libortk.so carries gfx950 code objects only")


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "libortk takes device pointers only"
    return C.c_void_p(t.data_ptr())


def check(code, what):
    if code != 0:
        names = {-1: "ORTK_EINVAL (bad argument / unsupported shape)", -2: "ORTK_ENOSPC (workspace too small)",
                 -3: "ORTK_ENOSYS (option not implemented)",
                 -4: "ORTK_EEXCHANGE (column-split decode: an exchange group never met — the launch was not fully resident; "
                     "outputs are poisoned; set model.exclusive_gpu = False when the GPU is shared)"}
        raise OrtkError(f"{what} failed: {names.get(code, 'hipError ' + str(code))}")


def f32c(t):
    """contiguous fp32 CUDA tensor (no copy when already so)"""
    assert t.is_cuda
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
