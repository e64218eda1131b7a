"""CPU-side checks (no GPU): the C-ABI library loads, exports every symbol include/ortk.h declares, the ctypes
table matches the header, the arena layout reproduces the reference's state_dict, and the product fails loudly
without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import common as Cm
import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ortk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ortk_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import sparse_image_captioning_amd as P
    lib = P._lib.lib()
    names = _header_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ortk.h but not exported by libortk.so"
    # the ctypes signature table covers the header one-to-one
    assert sorted(P._lib.SIGNATURES) == names
    assert lib.ortk_version() == P._lib.ABI_VERSION == int(re.search(r"#define ORTK_VERSION (\d+)", open(os.path.join(ROOT, "include", "ortk.h")).read()).group(1))


def test_every_header_under_include_is_fully_exported():
    """All C-ABI headers (device path, SCST scorer, batch padding): each declared ortk_* function is a symbol of libortk.so."""
    import glob
    import sparse_image_captioning_amd as P
    lib = P._lib.lib()
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert [os.path.basename(h) for h in headers] == ["ortk.h", "ortk_data.h", "ortk_scorer.h"]
    for h in headers:
        src = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names = sorted(set(re.findall(r"\b(ortk_[a-z0-9_]+)\s*\(", src)))
        assert names, h
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {os.path.basename(h)} but not exported by libortk.so"


def test_graft_entry_build_runs_and_checks_the_abi_version():
    """__graft_entry__.build() is the driver's "does it build" check: it must pass on the CPU box, and against the bindings' own
    ABI version (it once held a literal that an ORTK_VERSION bump left behind)."""
    import __graft_entry__ as G
    import sparse_image_captioning_amd as P
    G.build()
    assert P._lib.lib().ortk_version() == P._lib.ABI_VERSION


def test_struct_sizes_match_header():
    """sizeof() of the ctypes mirrors == what the C compiler lays out (checked by compiling a tiny C program)."""
    import subprocess, tempfile
    import sparse_image_captioning_amd as P
    prog = r'''
    #include <stdio.h>
    #include "ortk.h"
    int main(void){ printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ortk_config), sizeof(ortk_batch), sizeof(ortk_decode_opts),
                           sizeof(ortk_gemm_args), sizeof(ortk_attn_args), sizeof(ortk_sparse_block), sizeof(ortk_sparse_plan),
                           sizeof(ortk_spmm_args), sizeof(ortk_chain_unit), sizeof(ortk_chain_args), sizeof(ortk_bchain_args),
                           sizeof(ortk_tuning)); return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(prog)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    L = P._lib
    assert [int(x) for x in out] == [C.sizeof(L.Config), C.sizeof(L.Batch), C.sizeof(L.DecodeOpts), C.sizeof(L.GemmArgs),
                                     C.sizeof(L.AttnArgs), C.sizeof(L.EllBlock), C.sizeof(L.EllPlanStruct), C.sizeof(L.SpmmArgs),
                                     C.sizeof(L.ChainUnit), C.sizeof(L.ChainArgs), C.sizeof(L.BChainArgs), C.sizeof(L.Tuning)]


@pytest.mark.parametrize("cfg", [Cm.TINY_CFG, Cm.FULL_CFG])
def test_arena_layout_matches_reference_state_dict(cfg):
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.utils.config import Config
    model = P.get_model("relation_transformer")(Config(**cfg))
    sd = model.state_dict()
    want = H.dense_param_shapes(cfg)
    want["model.tgt_embed.1.pe"] = (1, 5000, cfg["d_model"])
    assert set(sd.keys()) == set(want.keys())
    for k, shp in want.items():
        assert tuple(sd[k].shape) == tuple(shp), k
    if cfg is Cm.FULL_CFG:
        assert sum(p.numel() for p in model.parameters()) == 55443777
        assert len(sd) == 358
    # every parameter is a view into ONE arena; fused blocks are adjacent
    base = model._flat.data_ptr()
    for p in model.parameters():
        assert base <= p.data_ptr() < base + model._flat.numel() * 4
    l0 = "model.encoder.layers.0.self_attn.linears."
    d = cfg["d_model"]
    q, k, v = (dict(model.named_parameters())[l0 + f"{i}.weight"] for i in range(3))
    assert k.data_ptr() - q.data_ptr() == d * d * 4 and v.data_ptr() - k.data_ptr() == d * d * 4
    # positional encoding equals the reference construction (transformer.py:369-374)
    from oracle import ort_oracle as O
    np.testing.assert_array_equal(sd["model.tgt_embed.1.pe"][0, :20].numpy(), O.positional_encoding(20, d).numpy())
    # xavier init under model.*, LayerNorm ones / zeros
    assert float(sd["model.decoder.norm.a_2"].min()) == 1.0 and float(sd["model.decoder.norm.b_2"].abs().max()) == 0.0
    w = sd["model.decoder.layers.0.feed_forward.w_1.weight"]
    bound = (6.0 / (w.shape[0] + w.shape[1])) ** 0.5
    assert float(w.abs().max()) <= bound * (1 + 1e-6) and float(w.abs().max()) > 0.9 * bound   # fp32 rounding of the bound


def test_state_dict_roundtrip_and_prune_keys():
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.utils.config import Config
    cfg = Config(**Cm.TINY_CFG)
    m = P.get_model("relation_transformer")(cfg)
    ref = H.g1_state()
    ref["model.tgt_embed.1.pe"] = m.state_dict()["model.tgt_embed.1.pe"].clone()
    m.load_state_dict(ref, strict=True)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k
    pm = P.get_model("relation_transformer_prune")(cfg)
    want = set(H.prune_param_shapes(Cm.TINY_CFG)) | {"model.tgt_embed.1.pe"}
    assert set(pm.state_dict().keys()) == want
    assert pm.mask_type == "supermask" and pm.total_mask_params == sum(
        int(np.prod(s)) for k, s in H.prune_param_shapes(Cm.TINY_CFG).items() if k.endswith("_pruning_mask"))
    assert all(float(m_.min()) == 5.0 for _, m_ in pm.all_pruning_masks())      # prune_supermask_init
    assert all(p.requires_grad for _, p in pm.all_pruning_masks())
    pm2 = P.get_model("relation_transformer_prune")(Config(**dict(Cm.TINY_CFG, prune_type="mag_blind")))
    assert all(float(m_.min()) == 1.0 and not m_.requires_grad for _, m_ in pm2.all_pruning_masks())
    with pytest.raises(ValueError):
        P.get_model("no_such_model")
    with pytest.raises(AssertionError):              # the reference's own `assert share_att in (None, "kv", "qk")`
        P.get_model("relation_transformer")(Config(**dict(Cm.TINY_CFG, share_att_encoder="vq")))
    for enc, dec in (("kv", "qk"), ("qk", "kv"), (None, "qk"), ("kv", None)):
        cfgd = dict(Cm.TINY_CFG, share_att_encoder=enc, share_att_decoder=dec)
        sm = P.get_model("relation_transformer_prune")(Config(**cfgd))
        assert set(sm.state_dict().keys()) == set(H.prune_param_shapes(cfgd)) | {"model.tgt_embed.1.pe"}
        # "qk" in the decoder: src_attn.linears.0 (query AND key projection) sits at the head of its layer's slice of the packed
        # cross-attention block, followed by linears.1 (value)
        off = {e["name"]: e["offset"] for e in sm._entries}
        d = cfgd["d_model"]
        if dec == "qk":
            assert off["model.decoder.layers.0.src_attn.linears.1.weight"] == off["model.decoder.layers.0.src_attn.linears.0.weight"] + d * d
            assert off["model.decoder.layers.1.src_attn.linears.0.weight"] == off["model.decoder.layers.0.src_attn.linears.0.weight"] + 2 * d * d


def test_prune_host_api_matches_golden(golden):
    """PruningMixin host logic (mask updates, statistics, sparse export) on CPU tensors vs the reference goldens."""
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.utils.config import Config
    g3 = golden("g3_tiny_prune")
    shapes = H.prune_param_shapes(Cm.TINY_CFG)
    for mtype in ("mag_blind", "mag_uniform", "mag_dist"):
        pm = P.get_model("relation_transformer_prune")(Config(**dict(Cm.TINY_CFG, prune_type=mtype)))
        sd = H.torch_state({k: v for k, v in shapes.items() if not k.endswith("_pruning_mask")}, Cm.G1_SEED,
                           Cm.G1_GEN_SCALE, Cm.G1_EOS_BIAS)
        pm.load_state_dict(sd, strict=False)
        pm.update_masks_once(0.8)
        names = g3[f"{mtype}/names"].tolist()
        ref = H.unpack_bits(g3[f"{mtype}/mask_bits"], [shapes[n] for n in names])
        masks = dict(pm.all_pruning_masks())
        assert sum(int((masks[n].detach().numpy() != r).sum()) for n, r in zip(names, ref)) == 0
        tot, nnz, per, _ = pm.all_mask_sparsities
        assert abs(float(tot) - float(g3[f"{mtype}/total"])) < 1e-6
    # supermask statistics + sparse / dense export round trip (prune.py:176-226, model_utils.py:110-118)
    pm = P.get_model("relation_transformer_prune")(Config(**Cm.TINY_CFG))
    sd = H.torch_state(shapes, Cm.G1_SEED, Cm.G1_GEN_SCALE, Cm.G1_EOS_BIAS, keep_prob=Cm.G3_KEEP)
    pm.load_state_dict(sd, strict=False)
    tot, nnz, per, names = pm.all_mask_sparsities
    assert abs(float(tot) - float(g3["sparsity/total"])) < 1e-6 and float(nnz) == float(g3["sparsity/nnz"])
    assert abs(float(pm.all_mask_avg) - float(g3["mask_avg"])) < 1e-5
    sparse = pm.state_dict_sparse()
    assert sum(int(v._nnz()) for v in sparse.values() if v.is_sparse) == int(g3["eval/sparse_nnz"])
    dense = P.get_model("relation_transformer")(Config(**Cm.TINY_CFG))
    dense.load_state_dict({k: (v.to_dense() if v.is_sparse else v) for k, v in sparse.items()}, strict=True)
    w = dict(dense.named_parameters())["model.generator.proj.weight"]
    m = torch.round(torch.sigmoid(sd["model.generator.proj.weight_pruning_mask"]))
    assert torch.equal(w.detach(), sd["model.generator.proj.weight"] * m)


def test_product_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.utils.config import Config
    m = P.get_model("relation_transformer")(Config(**Cm.TINY_CFG))
    b = H.g1_batch()
    with pytest.raises(P._lib.OrtkUnavailable):
        m(att_feats=b["att_feats"], boxes=b["boxes"], seqs=b["seqs"], att_masks=b["att_masks"])
    with pytest.raises(P._lib.OrtkUnavailable):
        m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], mode="sample")


def test_product_never_imports_the_oracle():
    """Grep-level guard: nothing under the package directory references oracle/ (judge rule ③)."""
    pkg = os.path.join(ROOT, "sparse-image-captioning_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "ort_oracle" not in txt.replace(
                    "oracle/ort_oracle.py", ""), os.path.join(dirpath, f)


def test_dim_mat_constants_match_torch():
    """The 8 wavelengths 1/1000^(k/8) used by the box kernel (computed with powf on the host) equal torch's fp32
    evaluation (relation_transformer.py:241-243) bit for bit."""
    k = torch.arange(8, dtype=torch.float32)
    ref = (1.0 / torch.pow(torch.tensor(1000.0), k / 8.0)).numpy()
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    mine = np.array([np.float32(1.0) / np.float32(libm.powf(1000.0, kk / 8.0)) for kk in range(8)], np.float32)
    np.testing.assert_array_equal(mine, ref)


def test_checkpoint_formats_sparse_coo_and_fp16_load_directly():
    """prune.py:200-221 / eval_model.py:64-88: a COO-sparse (and fp16-cast) checkpoint of the prune model loads into the dense
    class without a manual densify step and reproduces the masked weights."""
    import sparse_image_captioning_amd as P
    from sparse_image_captioning_amd.utils.config import Config
    from sparse_image_captioning_amd.utils.model_utils import densify_state_dict, count_nonzero
    cfg = Config(**Cm.TINY_CFG)
    pm = P.get_model("relation_transformer_prune")(cfg)
    pm.load_state_dict(H.torch_state(H.prune_param_shapes(Cm.TINY_CFG), Cm.G1_SEED, keep_prob=Cm.G3_KEEP), strict=False)
    sparse = pm.state_dict_sparse()
    assert any(v.is_sparse for v in sparse.values())
    half = {k: (v.half() if not v.is_sparse else v) for k, v in sparse.items()}
    dense = P.get_model("relation_transformer")(cfg)
    missing, unexpected = dense.load_state_dict(half, strict=False)
    assert not unexpected
    ref = densify_state_dict(sparse)
    got = dense.state_dict()
    for k, v in ref.items():
        assert torch.allclose(got[k], v.float(), atol=2e-3), k
    w = got["model.decoder.layers.0.feed_forward.w_1.weight"]
    assert 0.1 < float(count_nonzero(w)) / w.numel() < 0.5


def test_label_smoothing_criterion_vs_reference_golden():
    """utils/losses.py:46-77 (used when --label_smoothing > 0, scripts/train_transformer.py:33-34)."""
    import json
    sys_path = os.path.join(ROOT, "tests", "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_losses", os.path.join(sys_path, "make_golden_losses.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    from sparse_image_captioning_amd.utils.losses import LabelSmoothing
    gold = json.load(open(os.path.join(sys_path, "g7_label_smoothing.json")))
    for sm, want in gold.items():
        x, tgt, mask = mg.inputs()
        x.requires_grad_()
        loss = LabelSmoothing(smoothing=float(sm))(x, tgt, mask)
        grad, = torch.autograd.grad(loss, x)
        assert abs(float(loss) - want["loss"]) < 1e-5
        assert abs(float(grad.abs().sum()) - want["grad_abs_sum"]) < 1e-4
        assert abs(float(grad[0, 0, 5]) - want["grad_0_0_5"]) < 1e-6
