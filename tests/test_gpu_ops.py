"""Operator-level parity on a real MI355X: every C-ABI op against the oracle's fp32 restatement (torch CPU ops) on
the same seeded inputs.  Tolerances: fp32 MFMA path 1e-5 relative class (stated per test); bf16 MFMA path 2e-2."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import ort_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    import sparse_image_captioning_amd as P
    P._lib.require_gpu()
    return P._lib


_KEEP = []


def dev(t):
    """Copy to the GPU and keep the tensor alive: the C-ABI takes raw pointers and launches asynchronously, so a
    temporary that dies right after `.data_ptr()` could be recycled by the caching allocator under the kernel."""
    d = t.cuda().contiguous()
    _KEEP.append(d)
    if len(_KEEP) > 512:
        torch.cuda.synchronize()
        del _KEEP[:256]
    return d


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).float()


def gemm(L, A, B, M, N, K, ta=0, tb=0, prec=0, **kw):
    a = L.GemmArgs()
    Cout = kw.pop("C", None)
    if Cout is None:
        Cout = torch.full((M, N), float("nan"), device="cuda")
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), Cout.stride(0)
    a.M, a.N, a.K, a.transA, a.transB, a.precision = M, N, K, ta, tb, prec
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(a, k, v.data_ptr())
            if k == "resid":
                a.ldr = v.stride(0)
            if k == "gate":
                a.ldg = v.stride(0)
        else:
            setattr(a, k, v)
    L.check(L.lib().ortk_gemm(C.byref(a), L.stream_ptr()), "ortk_gemm")
    torch.cuda.synchronize()
    return Cout


@pytest.mark.parametrize("prec,tol", [(0, 2e-5), (1, 2e-2)])
@pytest.mark.parametrize("M,N,K", [(36, 64, 96), (300, 130, 70), (257, 10001 % 997 + 1, 512), (128, 128, 16), (1, 5, 3)])
def test_gemm_layouts(L, prec, tol, M, N, K):
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    ref = A @ B.t()
    scale = ref.abs().max().item() + 1e-6
    out = gemm(L, dev(A), dev(B), M, N, K, 0, 0, prec)                       # forward:  X W^T
    assert (out.cpu() - ref).abs().max().item() <= tol * scale
    out = gemm(L, dev(A), dev(B.t().contiguous()), M, N, K, 0, 1, prec)      # dgrad:    dY W   (B stored K x N)
    assert (out.cpu() - ref).abs().max().item() <= tol * scale
    out = gemm(L, dev(A.t().contiguous()), dev(B.t().contiguous()), M, N, K, 1, 1, prec)   # wgrad: both stored K-major
    assert (out.cpu() - ref).abs().max().item() <= tol * scale


@pytest.mark.parametrize("M,N,K", [(640, 384, 512), (5120, 512, 2048), (257, 1000, 544), (1, 5, 32), (5000, 10112, 512)])
def test_gemm_f32_split_vs_float64(L, M, N, K):
    """fp32 products of the forward layout run on the bf16 matrix cores: every operand split into three bf16 parts, six partial
    products kept (ortk_gemm.hip: gemm_f32x3_kernel; ortk_tuning.f32_split).  Held to the error of the fp32 MFMA kernel it replaces,
    both against a float64 product of the same fp32 operands: the split form's rms error may not exceed 1.25 x the native
    kernel's, its max error 2 x, and both stay inside 4 sqrt(K) 2^-24 of the result's scale.  Operands with a wide dynamic range
    (magnitudes over 2^+-20) and exact zeros included; every tile shape gives the same bits (same k order); epilogue as the
    native kernel's."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-20, 21, (M, K), generator=g).float())
    B = torch.randn(N, K, generator=g) * 0.05
    A[torch.rand(M, K, generator=g) < 0.1] = 0.0
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    Ad, Bd, bd, rd = dev(A), dev(B), dev(bias), dev(res)
    ref = A.double() @ B.double().t()
    scale = ref.abs().max().item()
    outs = {}
    prev = L.set_tuning(f32_split=0)
    try:
        for v in (0, 1, 2, 3, 4, 5, 6, 7):
            L.set_tuning(f32_split=v)
            outs[v] = gemm(L, Ad, Bd, M, N, K).cpu()
        L.set_tuning(f32_split=1)
        full = gemm(L, Ad, Bd, M, N, K, bias=bd, relu=1, resid=rd).cpu()
        L.set_tuning(f32_split=0)
        full0 = gemm(L, Ad, Bd, M, N, K, bias=bd, relu=1, resid=rd).cpu()
    finally:
        L.set_tuning(**prev)
    for v in (1, 3, 4, 5, 6, 7):
        assert torch.equal(outs[2], outs[v])
    e0, e1 = (outs[0].double() - ref), (outs[2].double() - ref)
    bar = 4 * math.sqrt(K) * 2.0 ** -24 * scale
    assert e0.abs().max().item() <= bar and e1.abs().max().item() <= bar
    if M * N >= 10000:      # (the comparison of two error samples needs a sample)
        assert e1.abs().max().item() <= 2.0 * e0.abs().max().item()
        assert e1.pow(2).mean().sqrt().item() <= 1.25 * e0.pow(2).mean().sqrt().item()
    torch.testing.assert_close(full, full0, rtol=0, atol=2 * bar)
    torch.testing.assert_close(full, (torch.relu(ref + bias.double()) + res.double()).float(), rtol=0, atol=2 * bar)


@pytest.mark.parametrize("M,N,K", [(640, 384, 512), (2048, 512, 5000), (260, 1000, 96), (4, 8, 33)])
def test_gemm_f32_split_transposed_layouts_vs_float64(L, M, N, K):
    """The data-gradient (B stored (K, N)) and weight-gradient (both operands stored k-major; split-K accumulation with atomics)
    layouts of the fp32 split product (ortk_gemm.hip: gemm_f32x3t_kernel) against a float64 product, next to the fp32 MFMA kernel
    they replace (f32_split = 0): same bars as the forward layout's test.  K that is no multiple of 32 (zero-filled k-rows), tiles
    that overhang M and N, bias + ReLU + residual epilogue, accumulation on top of existing content."""
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-12, 13, (M, K), generator=g).float())
    B = torch.randn(N, K, generator=g) * 0.05
    ref = A.double() @ B.double().t()
    scale = ref.abs().max().item()
    bar = 4 * math.sqrt(K) * 2.0 ** -24 * scale
    Ad, Bd = dev(A), dev(B)
    Atd, Btd = dev(A.t().contiguous()), dev(B.t().contiguous())
    C0 = rnd(M, N, seed=11)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    out = {}
    prev = L.set_tuning(f32_split=0)
    try:
        for v in (0, 1):
            L.set_tuning(f32_split=v)
            o = {}
            if K % 32 == 0:
                o["dgrad"] = gemm(L, Ad, Btd, M, N, K, 0, 1).cpu()
                o["dgrad_epi"] = gemm(L, Ad, Btd, M, N, K, 0, 1, bias=dev(bias), relu=1, resid=dev(res)).cpu()
            csum = torch.zeros(M, device="cuda")
            o["wgrad"] = gemm(L, Atd, Btd, M, N, K, 1, 1, colsum=csum).cpu()      # + fused column sums of A (the bias gradient)
            o["colsum"] = csum.cpu()
            for sk in (1, 3, 8):
                o["wgrad_acc%d" % sk] = gemm(L, Atd, Btd, M, N, K, 1, 1, C=dev(C0.clone()), accumulate=1, splitk=sk).cpu()
            out[v] = o
    finally:
        L.set_tuning(**prev)
    for v in (0, 1):
        cs = out[v].pop("colsum")
        torch.testing.assert_close(cs.double(), A.double().sum(1), rtol=0, atol=4 * math.sqrt(K) * 2.0 ** -24 * A.abs().sum(1).max().item())
    for name in out[1]:
        want = ref
        if name == "dgrad_epi":
            want = torch.relu(ref + bias.double()) + res.double()
        elif name.startswith("wgrad_acc"):
            want = ref + C0.double()
        e0, e1 = out[0][name].double() - want, out[1][name].double() - want
        assert e1.abs().max().item() <= 2 * bar, name
        if M * N >= 10000:
            assert e1.pow(2).mean().sqrt().item() <= 1.25 * e0.pow(2).mean().sqrt().item() + 2.0 ** -24 * scale, name


def test_gemm_epilogues_and_splitk(L):
    M, N, K = 200, 72, 160
    A, B, bias, res, rs = rnd(M, K, seed=3), rnd(N, K, seed=4), rnd(N, seed=5), rnd(M, N, seed=6), (rnd(M, seed=7) > 0).float()
    out = gemm(L, dev(A), dev(B), M, N, K, bias=dev(bias), relu=1, rowscale=dev(rs), resid=dev(res))
    ref = torch.relu(A @ B.t() + bias) * rs[:, None] + res
    torch.testing.assert_close(out.cpu(), ref, rtol=2e-5, atol=2e-4)
    gate = rnd(M, N, seed=8)
    out = gemm(L, dev(A), dev(B), M, N, K, gate=dev(gate), gate_scale=1.25)
    torch.testing.assert_close(out.cpu(), (A @ B.t()) * (gate > 0).float() * 1.25, rtol=2e-5, atol=2e-4)
    # accumulate with split-K (wgrad form): C += A^T B over 8 K-slices
    Kb = 2048
    A2, B2 = rnd(Kb, M, seed=9), rnd(Kb, N, seed=10)
    C0 = rnd(M, N, seed=11)
    out = gemm(L, dev(A2), dev(B2), M, N, Kb, 1, 1, C=dev(C0.clone()), accumulate=1, splitk=8)
    torch.testing.assert_close(out.cpu(), C0 + A2.t() @ B2, rtol=1e-4, atol=2e-3)
    # dropout epilogue: keep-rate and scaling
    out = gemm(L, dev(A), dev(B), M, N, K, drop_p=0.25, drop_seed=1234)
    ref = A @ B.t()
    kept = out.cpu() != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.02
    torch.testing.assert_close(out.cpu()[kept], (ref / 0.75)[kept], rtol=2e-5, atol=2e-4)
    out2 = gemm(L, dev(A), dev(B), M, N, K, drop_p=0.25, drop_seed=1234)
    assert torch.equal(out, out2)                                            # counter-based: reproducible


@pytest.mark.parametrize("mode", ["atomics", "workspace"])
@pytest.mark.parametrize("rows,shapes,splitk,ldpad", [
    (1017, [(512, 512), (1536, 512), (512, 2048), (2048, 512)], 0, 0),        # a layer's projections, ragged row count
    (333, [(264, 520), (8, 8), (128, 1000)], 0, 8),                            # partial tiles, padded leading dimensions
    (4097, [(10112, 512)], 3, 0),                                              # the generator's padded vocabulary, three row ranges
    (31, [(512, 256)], 0, 0), (32, [(512, 256)], 0, 0), (69, [(256, 256)], 0, 0),   # fewer rows than a stage; exactly one; one + a tail
    (16640, [(512, 512), (1536, 512)], 4, 0),                                  # the XE step's valid decoder rows, four row ranges
    (2048, [(512, 512)] * 8, 2, 0)])                                           # ORTK_WGRAD_MAX projections
def test_wgrad_group_vs_torch(L, mode, rows, shapes, splitk, ldpad):
    """ortk_wgrad_group: dW_i += dY_i^T X_i and db_i += colsum(dY_i) of every projection of the group in one launch, against torch
    fp32 products of the same bf16 operands (the per-Linear weight / bias autograd of the reference's step,
    scripts/train_transformer.py:65-81).  fp32 accumulation of exact bf16 products: the bar is the order of the additions
    (1e-5 of the largest entry), in both reduction forms (atomic rows / partial tiles + last arriver) and on a reused workspace."""
    lib = L.lib()
    a = L.WgradGroupArgs(); a.n = len(shapes); a.rows = rows; a.splitk = splitk
    items = []
    for i, (n, k) in enumerate(shapes):
        dY = dev((rnd(rows, n + ldpad, seed=3 * i) * 0.5).bfloat16()); X = dev(rnd(rows, k + ldpad, seed=3 * i + 1).bfloat16())
        dW = dev(rnd(n, k, seed=3 * i + 2)); db = dev(rnd(n, seed=3 * i + 5))
        it = a.item[i]
        it.dY, it.lddy, it.X, it.ldx, it.dW, it.lddw, it.db, it.Nout, it.Kin = dY.data_ptr(), n + ldpad, X.data_ptr(), k + ldpad, dW.data_ptr(), k, db.data_ptr(), n, k
        items.append((dY, X, dW, db, n, k, dW.clone(), db.clone()))
    need = lib.ortk_wgrad_group_workspace_bytes(C.byref(a))
    if mode == "workspace":
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device="cuda"); _KEEP.append(ws)
        a.ws, a.ws_bytes = ws.data_ptr(), need
    for rep in range(2):          # the second launch adds on top of the first (and reuses the workspace)
        L.check(lib.ortk_wgrad_group(C.byref(a), L.stream_ptr()), "ortk_wgrad_group")
    torch.cuda.synchronize()
    for dY, X, dW, db, n, k, dW0, db0 in items:
        ref = dW0.double() + 2 * (dY[:, :n].double().t() @ X[:, :k].double())
        refb = db0.double() + 2 * dY[:, :n].double().sum(0)
        assert (dW.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
        assert (db.double() - refb).abs().max().item() <= 1e-5 * refb.abs().max().item()
    # what the launcher refuses: odd widths, misaligned operands, a workspace that is too small
    bad = L.WgradGroupArgs.from_buffer_copy(a); bad.item[0].Nout = shapes[0][0] - 1
    assert lib.ortk_wgrad_group(C.byref(bad), L.stream_ptr()) == -1
    bad = L.WgradGroupArgs.from_buffer_copy(a); bad.item[0].dY = a.item[0].dY + 2
    assert lib.ortk_wgrad_group(C.byref(bad), L.stream_ptr()) == -1
    if mode == "workspace" and need:
        bad = L.WgradGroupArgs.from_buffer_copy(a); bad.ws_bytes = need - 1
        assert lib.ortk_wgrad_group(C.byref(bad), L.stream_ptr()) == -2


@pytest.mark.parametrize("rows,d", [(7, 64), (300, 512), (33, 100), (5, 2048)])
def test_layernorm_fwd_bwd(L, rows, d):
    x, a, b, dy, dres = rnd(rows, d, seed=1, scale=2.0), 1 + 0.1 * rnd(d, seed=2), 0.1 * rnd(d, seed=3), rnd(rows, d, seed=4), rnd(rows, d, seed=5)
    xr, ar, br = x.clone().requires_grad_(), a.clone().requires_grad_(), b.clone().requires_grad_()
    ref = O.layer_norm(xr, ar, br)
    ref.backward(dy)
    y = torch.empty(rows, d, device="cuda"); st = torch.empty(rows, 2, device="cuda")
    xd, ad, bd = dev(x), dev(a), dev(b)
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(xd), L.ptr(ad), L.ptr(bd), L.ptr(y), 0, L.ptr(st), rows, d, 1e-6, L.stream_ptr()), "ln")
    torch.testing.assert_close(y.cpu(), ref.detach(), rtol=1e-5, atol=1e-5)
    dx = torch.empty(rows, d, device="cuda"); da = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda")
    L.check(L.lib().ortk_layernorm_bwd(L.ptr(dev(dy)), L.ptr(xd), L.ptr(ad), L.ptr(st), L.ptr(dev(dres)), L.ptr(dx), L.ptr(da),
                                       L.ptr(db), rows, d, 1e-6, L.stream_ptr()), "ln_bwd")
    torch.testing.assert_close(dx.cpu(), xr.grad + dres, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(da.cpu(), ar.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.cpu(), br.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("rows,d", [(16640, 512), (333, 512), (100, 256)])
def test_layernorm_bwd_element_types_and_forms(L, rows, d):
    """ortk_layernorm_bwd_dt on a bf16 output gradient == the fp32 entry on the same (bf16-rounded) values; the eight-column form of
    width 512 == the four-column form (ortk_tuning.ln_fuse bit 2) up to the order of the row sums; both with the residual gradient, the
    dropout-masked bf16 copy and the parameter gradients of the executor's call."""
    lib = L.lib()
    x, dy, a, b, dres = dev(rnd(rows, d, seed=1)), rnd(rows, d, seed=2), dev(rnd(d, seed=3)), dev(rnd(d, seed=4)), dev(rnd(rows, d, seed=5))
    dy16 = dev(dy.bfloat16()); dy32 = dy16.float()
    y = torch.empty(rows, d, device="cuda"); st = torch.empty(rows, 2, device="cuda")
    L.check(lib.ortk_layernorm_fwd(L.ptr(x), L.ptr(a), L.ptr(b), L.ptr(y), 0, L.ptr(st), rows, d, 1e-6, L.stream_ptr()), "fwd")
    out = {}
    try:
        for key, g, dt, fuse in (("f32", dy32, 0, 0), ("bf16", dy16, 1, 0), ("four", dy32, 0, 4)):
            L.set_tuning(ln_fuse=fuse)
            dx = torch.full((rows, d), float("nan"), device="cuda"); dz = torch.zeros(rows, d, device="cuda", dtype=torch.bfloat16)
            da, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
            rc = lib.ortk_layernorm_bwd_dt(L.ptr(g), dt, L.ptr(x), L.ptr(a), L.ptr(st), L.ptr(dres), L.ptr(dx), L.ptr(da), L.ptr(db), rows, d, 1e-6,
                                           L.ptr(dz), 1, 0.1, 7, None, L.stream_ptr())
            if dt == 1 and d != 512:          # bf16 output gradients: the model width only (include/ortk.h) — refused, not mis-read
                assert rc == -1
                continue
            L.check(rc, "bwd")
            torch.cuda.synchronize()
            out[key] = (dx, dz, da, db)
    finally:
        L.set_tuning(ln_fuse=0)
    sc = out["f32"][0].abs().max().item()
    for key in ("bf16", "four"):
        if key not in out:
            continue
        assert (out[key][0] - out["f32"][0]).abs().max().item() <= 4e-6 * sc, key
        assert ((out[key][1].float() != 0) == (out["f32"][1].float() != 0)).all(), key                # the same dropout mask
        assert (out[key][1].float() - out["f32"][1].float()).abs().max().item() <= 2 ** -7 * sc, key      # (bf16 copies: one rounding apart at most)
        for u in (2, 3):
            torch.testing.assert_close(out[key][u], out["f32"][u], rtol=1e-4, atol=1e-3 * rows ** 0.5)


@pytest.mark.parametrize("rows,d,p,dt", [(300, 512, 0.1, 1), (77, 512, 0.0, 1), (33, 100, 0.25, 0), (64, 2048, 0.1, 1)])
def test_layernorm_bwd_with_fused_dropout_output(L, rows, d, p, dt):
    """ortk_layernorm_bwd_drop == ortk_layernorm_bwd followed by ortk_dropout_apply on its dx (bitwise)."""
    x, dy, a, b, dres = rnd(rows, d, seed=1), rnd(rows, d, seed=2), 1 + 0.1 * rnd(d, seed=3), rnd(d, seed=4), rnd(rows, d, seed=5)
    xd, dyd, ad, bd, rd = dev(x), dev(dy), dev(a), dev(b), dev(dres)
    y = torch.empty(rows, d, device="cuda"); st = torch.empty(rows, 2, device="cuda")
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(xd), L.ptr(ad), L.ptr(bd), L.ptr(y), 0, L.ptr(st), rows, d, 1e-6, L.stream_ptr()), "ln")
    tdt = torch.bfloat16 if dt else torch.float32
    dx1, dx2 = torch.empty_like(xd), torch.empty_like(xd)
    z1, z2 = torch.empty(rows, d, device="cuda", dtype=tdt), torch.empty(rows, d, device="cuda", dtype=tdt)
    da1, db1, da2, db2 = (torch.zeros(d, device="cuda") for _ in range(4))
    L.check(L.lib().ortk_layernorm_bwd(L.ptr(dyd), L.ptr(xd), L.ptr(ad), L.ptr(st), L.ptr(rd), L.ptr(dx1), L.ptr(da1), L.ptr(db1),
                                       rows, d, 1e-6, L.stream_ptr()), "ln_bwd")
    L.check(L.lib().ortk_dropout_apply(L.ptr(dx1), L.ptr(z1), dt, rows * d, p, 99, L.stream_ptr()), "drop")
    L.check(L.lib().ortk_layernorm_bwd_drop(L.ptr(dyd), L.ptr(xd), L.ptr(ad), L.ptr(st), L.ptr(rd), L.ptr(dx2), L.ptr(da2), L.ptr(db2),
                                            rows, d, 1e-6, L.ptr(z2), dt, p, 99, L.stream_ptr()), "ln_bwd_drop")
    assert torch.equal(dx1, dx2) and torch.equal(z1, z2)
    torch.testing.assert_close(da1, da2, rtol=1e-5, atol=1e-5)
    if p > 0:
        frac = (z2 == 0).float().mean().item()
        assert abs(frac - p) < 0.03


@pytest.mark.parametrize("R,T", [(1, 1), (5, 17), (1280, 18), (5000, 64), (65536, 3)])
def test_valid_position_tables(L, R, T):
    """ortk_valid_position_tables: cap_off = exclusive prefix sums of the per-caption position counts, row_pos[cap_off[r] + t] =
    r * T + t — the tables of the valid-position decoder layout (ortk_batch.cap_off / row_pos), built on the device."""
    g = torch.Generator().manual_seed(R + T)
    n = torch.randint(1, T + 1, (R,), generator=g)
    Mc = int(n.sum())
    off = torch.full((R + 1,), -1, dtype=torch.int32, device="cuda")
    rows = torch.full((Mc,), -1, dtype=torch.int32, device="cuda")
    L.check(L.lib().ortk_valid_position_tables(L.ptr(dev(n)), R, T, L.ptr(off), L.ptr(rows), L.stream_ptr()), "tables")
    ref_off = torch.zeros(R + 1, dtype=torch.int64); ref_off[1:] = torch.cumsum(n, 0)
    ref_rows = torch.repeat_interleave(torch.arange(R) * T - ref_off[:-1], n) + torch.arange(Mc)
    assert torch.equal(off.cpu().long(), ref_off) and torch.equal(rows.cpu().long(), ref_rows)


def test_dropout_draws_keyed_by_a_row_map(L):
    """`drop_rows` (ortk_gemm_args / ortk_spmm_args / ortk_layernorm_bwd_drop_rows / ortk_dropout_apply_rows): output row m takes
    the draws of row drop_rows[m] — how the valid-position decoder layout (ortk_batch.row_pos) draws what the padded (caption,
    position) layout draws.  Property, per operator: on a subset `sel` of the rows with drop_rows = sel, the operator returns
    exactly the rows `sel` of its result on all rows (bit for bit: same products, same draws)."""
    from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
    g = torch.Generator().manual_seed(5)
    Mfull = 1536
    sel = torch.randperm(Mfull, generator=g)[:1000].sort().values
    seld, selc = dev(sel.to(torch.int32)), sel.cuda()
    M = sel.numel()
    # GEMM epilogues: fp32 kernel (guarded), bf16 register-staged kernel (ragged N), bf16 LDS-DMA kernels (full tiles)
    for prec, N, K, bf in [(0, 130, 70, False), (1, 130, 72, True), (1, 512, 512, True), (1, 2048, 512, True)]:
        A, B, bias, res = rnd(Mfull, K, seed=1), rnd(N, K, seed=2), rnd(N, seed=3), rnd(Mfull, N, seed=4)
        Ad, Bd = dev(A.bfloat16() if bf else A), dev(B.bfloat16() if bf else B)
        dt = dict(a_dtype=1, b_dtype=1) if bf else {}
        full = gemm(L, Ad, Bd, Mfull, N, K, 0, 0, prec, bias=dev(bias), resid=dev(res), drop_p=0.3, drop_seed=77, **dt)
        As, Rs = Ad[selc].contiguous(), dev(res)[selc].contiguous()
        part = gemm(L, As, Bd, M, N, K, 0, 0, prec, bias=dev(bias), resid=Rs, drop_p=0.3, drop_seed=77, drop_rows=seld, **dt)
        assert torch.equal(part, full[selc]), (prec, N, K)
        plain = gemm(L, As, Bd, M, N, K, 0, 0, prec, bias=dev(bias), resid=Rs, drop_p=0.3, drop_seed=77, **dt)
        assert not torch.equal(plain, part)                      # (without the map the subset draws as rows 0 .. M - 1)
    # residual-branch dropout of a gradient, alone and fused into the LayerNorm backward
    d = 512
    x, dy, a_, b_ = rnd(Mfull, d, seed=6), rnd(Mfull, d, seed=7), 1 + 0.1 * rnd(d, seed=8), rnd(d, seed=9)
    xd, dyd, ad, bd = dev(x), dev(dy), dev(a_), dev(b_)
    zf = torch.empty(Mfull, d, device="cuda", dtype=torch.bfloat16)
    L.check(L.lib().ortk_dropout_apply(L.ptr(dyd), L.ptr(zf), 1, Mfull * d, 0.2, 31, L.stream_ptr()), "drop")
    zp = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    L.check(L.lib().ortk_dropout_apply_rows(L.ptr(dyd[selc].contiguous()), L.ptr(zp), 1, M, d, 0.2, 31, L.ptr(seld), L.stream_ptr()), "drop_rows")
    assert torch.equal(zp, zf[selc])
    y = torch.empty(Mfull, d, device="cuda"); st = torch.empty(Mfull, 2, device="cuda")
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(xd), L.ptr(ad), L.ptr(bd), L.ptr(y), 0, L.ptr(st), Mfull, d, 1e-6, L.stream_ptr()), "ln")

    def ln_bwd(xx, dd, ss, rows, key):
        dx = torch.empty(rows, d, device="cuda"); z = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16)
        da, db = torch.zeros(d, device="cuda"), torch.zeros(d, device="cuda")
        L.check(L.lib().ortk_layernorm_bwd_drop_rows(L.ptr(dd), L.ptr(xx), L.ptr(ad), L.ptr(ss), None, L.ptr(dx), L.ptr(da), L.ptr(db), rows, d,
                                                     1e-6, L.ptr(z), 1, 0.2, 32, L.ptr(key) if key is not None else None, L.stream_ptr()), "ln_bwd_drop_rows")
        return dx, z

    dxf, zf2 = ln_bwd(xd, dyd, st, Mfull, None)
    dxp, zp2 = ln_bwd(xd[selc].contiguous(), dyd[selc].contiguous(), st[selc].contiguous(), M, seld)
    assert torch.equal(dxp, dxf[selc]) and torch.equal(zp2, zf2[selc])
    # sparse products (the masked linears as ELL / GU images): the GEMM's epilogue, the GEMM's keys
    N, K = 512, 512
    W = rnd(N, K, seed=11, scale=0.2) * (torch.rand(N, K, generator=g) >= 0.9).float()
    X = dev(rnd(Mfull, K, seed=12).bfloat16())
    for fmt in (L.SP_ELL16, L.SP_GU16):
        plan = SparsePlan([dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 0.2))], fmt, "cuda")
        plan.build(dev(W.bfloat16())); plan.check_overflow()

        def run(Xr, rows, key):
            Y = torch.empty(rows, N, device="cuda")
            a = L.SpmmArgs()
            a.X, a.Y, a.ldx, a.ldy, a.M, a.x_dtype, a.y_dtype = Xr.data_ptr(), Y.data_ptr(), K, N, rows, 1, 0
            a.drop_p, a.drop_seed = 0.3, 55
            if key is not None:
                a.drop_rows = key.data_ptr()
            plan.spmm(0, a)
            torch.cuda.synchronize()
            return Y

        yf = run(X, Mfull, None)
        yp = run(X[selc].contiguous(), M, seld)
        assert torch.equal(yp, yf[selc]), fmt


def _boxes(B, S, seed):
    import common as Cm
    return torch.from_numpy(Cm.make_inputs(seed, B, S, 4, 10, 1)["boxes"])


def test_box_embedding_and_logbias(L):
    B, S, H, Lyr = 3, 13, 8, 2
    boxes = _boxes(B, S, 5)
    dm = (1.0 / torch.pow(torch.tensor(1000.0), torch.arange(8.0) / 8.0)).numpy().astype(np.float32)
    dmc = (C.c_float * 8)(*dm.tolist())
    emb = torch.empty(B, S, S, 64, device="cuda")
    L.check(L.lib().ortk_box_embedding(L.ptr(dev(boxes)), dmc, L.ptr(emb), B, S, L.stream_ptr()), "emb")
    ref = O.box_relational_embedding(boxes)
    # arguments reach ~690 rad: 1 ulp of the fp32 argument moves sin/cos by 6e-5
    assert (emb.cpu() - ref).abs().max().item() < 2e-4
    wg = [rnd(H, 64, seed=10 + l, scale=0.3) for l in range(Lyr)]
    bg = [rnd(H, seed=20 + l, scale=0.1) + 0.3 for l in range(Lyr)]
    wgd, bgd = [dev(w) for w in wg], [dev(b) for b in bg]
    PP = C.c_void_p * Lyr
    out = torch.empty(Lyr, B, H, S, S, device="cuda")
    L.check(L.lib().ortk_box_logbias_fwd(L.ptr(dev(boxes)), PP(*[w.data_ptr() for w in wgd]), PP(*[b.data_ptr() for b in bgd]),
                                         dmc, L.ptr(out), Lyr, B, S, H, L.stream_ptr()), "logbias")
    wr, br = [w.clone().requires_grad_() for w in wg], [b.clone().requires_grad_() for b in bg]
    refs = []
    for l in range(Lyr):
        g = torch.relu(torch.einsum("bijk,hk->bhij", ref, wr[l]) + br[l][None, :, None, None])
        refs.append(torch.log(torch.clamp(g, min=1e-6)))
    refs = torch.stack(refs)
    assert (out.cpu().exp() - refs.detach().exp()).abs().max().item() < 2e-4
    dscore = rnd(Lyr, B, H, S, S, seed=30)
    refs.backward(dscore)
    dwg = [torch.zeros(H, 64, device="cuda") for _ in range(Lyr)]
    dbg = [torch.zeros(H, device="cuda") for _ in range(Lyr)]
    L.check(L.lib().ortk_box_logbias_bwd(L.ptr(dev(boxes)), PP(*[w.data_ptr() for w in wgd]), PP(*[b.data_ptr() for b in bgd]),
                                         dmc, L.ptr(dev(dscore)), PP(*[w.data_ptr() for w in dwg]), PP(*[b.data_ptr() for b in dbg]),
                                         Lyr, B, S, H, L.stream_ptr()), "logbias_bwd")
    for l in range(Lyr):
        # 1/pre amplifies the sin/cos argument noise where pre is tiny: compare at 2 % of the gradient scale
        sc = wr[l].grad.abs().max().item()
        assert (dwg[l].cpu() - wr[l].grad).abs().max().item() < 2e-2 * sc
        assert (dbg[l].cpu() - br[l].grad).abs().max().item() < 2e-2 * br[l].grad.abs().max().item()


def _attn_case(L, nkv, H, Lq, Lk, dk, causal, use_bias, use_mask, seed=0, precision=0, out_dt=0, in_dt=0):
    d = H * dk
    q, k, v = rnd(nkv * Lq, d, seed=seed + 1), rnd(nkv * Lk, d, seed=seed + 2), rnd(nkv * Lk, d, seed=seed + 3)
    kmask = torch.ones(nkv, Lk)
    if use_mask:
        for g in range(nkv):
            kmask[g, max(1, Lk - 1 - g % Lk):] = 0
    bias = rnd(nkv, H, Lq, Lk, seed=seed + 4) if use_bias else None
    do = rnd(nkv * Lq, d, seed=seed + 5)
    qr, kr, vr = q.clone().requires_grad_(), k.clone().requires_grad_(), v.clone().requires_grad_()
    br = bias.clone().requires_grad_() if use_bias else None
    qh = qr.view(nkv, Lq, H, dk).transpose(1, 2); kh = kr.view(nkv, Lk, H, dk).transpose(1, 2); vh = vr.view(nkv, Lk, H, dk).transpose(1, 2)
    mask = kmask[:, None, None, :].bool()
    if causal:
        qpos = torch.arange(Lq) % causal
        mask = mask & (torch.arange(Lk)[None, :] <= qpos[:, None])[None, None]
    ref = O.attention(qh, kh, vh, mask, br)
    ref_o = ref.transpose(1, 2).reshape(nkv * Lq, d)
    ref_o.backward(do)
    a = L.AttnArgs()
    a.precision = precision; a.o_dtype = a.dqkv_dtype = out_dt
    odt = torch.bfloat16 if out_dt else torch.float32
    # bf16 operands: error relative to the tensor's scale (the fp32 kernels keep the tight element-wise bound)
    def close(x, y, rtol, atol):
        if precision:
            err = (x.float() - y).abs().max().item()
            assert err < 2e-2 * max(y.abs().max().item(), 1e-3), err
        else:
            torch.testing.assert_close(x, y, rtol=rtol, atol=atol)
    qd, kd, vd, dod = dev(q), dev(k), dev(v), dev(do)
    if in_dt:       # bf16 Q / K / V rows (the executor's packed projections in mixed precision)
        qd, kd, vd, dod = qd.bfloat16(), kd.bfloat16(), vd.bfloat16(), dod.bfloat16()      # ... and the backward's dO
        a.qkv_dtype = 1
    o = torch.empty(nkv * Lq, d, device="cuda", dtype=odt); p = torch.empty(nkv, H, Lq, Lk, device="cuda")
    a.q, a.k, a.v, a.o = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr()
    a.ldq = a.ldk = a.ldv = a.ldo = d
    km = dev(kmask); a.kmask = km.data_ptr()
    if use_bias:
        bd = dev(bias); a.bias = bd.data_ptr()
    a.p = p.data_ptr(); a.nkv, a.H, a.Lq, a.Lk, a.dk, a.causal_period = nkv, H, Lq, Lk, dk, causal
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    close(o.cpu(), ref_o.detach(), 1e-4, 1e-5)
    dq, dk_, dv = (torch.empty_like(t, dtype=odt) for t in (qd, kd, vd))
    ds = torch.empty(nkv, H, Lq, Lk, device="cuda")
    a.d_o, a.dq, a.d_k, a.dv, a.dscore = dod.data_ptr(), dq.data_ptr(), dk_.data_ptr(), dv.data_ptr(), ds.data_ptr()
    a.lddo = a.lddq = a.lddk = a.lddv = d
    L.check(L.lib().ortk_attention_bwd(C.byref(a), L.stream_ptr()), "attn_bwd")
    close(dq.cpu(), qr.grad, 1e-4, 2e-5)
    close(dk_.cpu(), kr.grad, 1e-4, 2e-5)
    close(dv.cpu(), vr.grad, 1e-4, 2e-5)
    if use_bias:
        close(ds.cpu(), br.grad, 1e-4, 2e-5)


def test_attention_shapes(L):
    _attn_case(L, 5, 8, 36, 36, 64, 0, True, True)       # encoder box attention
    _attn_case(L, 7, 8, 17, 17, 64, 17, False, True)     # decoder self attention (causal + pad)
    _attn_case(L, 3, 8, 85, 36, 64, 0, False, True)      # cross attention: 5 captions x 17 rows share an image's K/V
    _attn_case(L, 4, 8, 12, 12, 8, 0, True, True)        # tiny golden geometry (dk = 8)
    _attn_case(L, 2, 2, 5, 100, 32, 0, False, True)      # 100 regions (2 keys per lane)
    _attn_case(L, 3, 4, 1, 1, 16, 1, False, False)       # single key
    _attn_case(L, 6, 3, 20, 9, 64, 0, True, True)        # register-only kernel: 2 query tiles x 1 key tile, bias
    _attn_case(L, 5, 2, 12, 31, 64, 0, False, True)      # register-only kernel: 1 x 2 tiles
    _attn_case(L, 9, 8, 32, 32, 64, 8, False, False)     # register-only kernel: full 2 x 2 tiles, causal period 8
    _attn_case(L, 6, 8, 5, 36, 64, 0, False, True)       # decode cross-attention: 5 beams x 36 regions (3 key tiles)
    _attn_case(L, 6, 8, 1, 36, 64, 0, False, True)       # first beam pass: one row per image
    _attn_case(L, 40, 8, 1, 13, 64, 0, False, False)     # decode self-attention: one row, all 8 heads in one wave
    _attn_case(L, 7, 8, 1, 32, 64, 0, False, True)       # ... longest supported cache, with key mask


def test_attention_bf16_operand_kernels(L):
    """ortk_attn_args.precision = 1: the bf16-MFMA block kernels (ortk_attn16.hip) on the training shapes, against the fp32
    torch attention at bf16-operand tolerance (2 % of each tensor's scale); fp32 and bf16 outputs."""
    _attn_case(L, 5, 8, 36, 36, 64, 0, True, True, precision=1)              # encoder box attention: 3 key tiles, k range padded to 64
    _attn_case(L, 3, 8, 85, 36, 64, 0, False, True, precision=1, out_dt=1)   # cross attention, bf16 O / dQ / dK / dV as in the executor
    _attn_case(L, 2, 4, 50, 64, 64, 0, True, True, precision=1)              # 4 full key tiles
    _attn_case(L, 3, 2, 40, 13, 64, 0, False, True, precision=1)             # one key tile, Lk not a multiple of 4 (scalar P rows)
    _attn_case(L, 2, 8, 51, 17, 64, 17, False, True, precision=1)            # causal period
    _attn_case(L, 2, 1, 128, 30, 64, 0, True, False, precision=1, out_dt=1)  # 8 waves
    # bf16 Q / K / V in memory (qkv_dtype = 1): the three stacks of the training step
    _attn_case(L, 5, 8, 36, 36, 64, 0, True, True, precision=1, out_dt=1, in_dt=1)
    _attn_case(L, 3, 8, 85, 36, 64, 0, False, True, precision=1, out_dt=1, in_dt=1)
    _attn_case(L, 7, 8, 17, 17, 64, 17, False, True, precision=1, out_dt=1, in_dt=1)     # decoder self-attention: one 16-row tile + 1 row
    _attn_case(L, 4, 2, 9, 5, 64, 9, False, False, precision=1, in_dt=1)                  # a single partial tile
    # 65-128 keys (ragged region counts up to 100 per image): wider [query][key] images, 3-4 k-steps over the keys
    _attn_case(L, 2, 8, 100, 100, 64, 0, True, True, precision=1, out_dt=1, in_dt=1)     # encoder self-attention with 100 regions
    _attn_case(L, 2, 8, 85, 100, 64, 0, False, True, precision=1, out_dt=1, in_dt=1)     # cross attention over 100 regions
    _attn_case(L, 1, 2, 128, 128, 64, 0, True, False, precision=1, in_dt=1)               # the largest shape
    _attn_case(L, 2, 4, 70, 77, 64, 0, True, True, precision=1)                           # fp32 inputs, Lk not a multiple of 4
    _attn_case(L, 6, 8, 5, 60, 64, 0, False, True, precision=1)                           # decode cross-attention: 5 beams x 60 regions (fp32 rows)
    # 32-wide heads (d_model 256 with 8 heads: the ACORT-small width)
    _attn_case(L, 3, 8, 36, 36, 32, 0, True, True, precision=1, out_dt=1, in_dt=1)
    _attn_case(L, 2, 8, 85, 36, 32, 0, False, True, precision=1, out_dt=1, in_dt=1)
    _attn_case(L, 5, 8, 25, 25, 32, 25, False, True, precision=1, out_dt=1, in_dt=1)      # causal, T = 25 (max_seq_length 26)
    _attn_case(L, 2, 4, 100, 100, 32, 0, True, True, precision=1)                         # fp32 rows, 100 keys
    a = L.AttnArgs()                                                                      # shapes the bf16 kernels do not take: loud
    a.qkv_dtype, a.precision = 1, 1
    t = torch.zeros(4 * 200, 64, device="cuda", dtype=torch.bfloat16); o = torch.zeros(4 * 200, 64, device="cuda")
    a.q = a.k = a.v = t.data_ptr(); a.o = o.data_ptr(); a.ldq = a.ldk = a.ldv = a.ldo = 64
    a.nkv, a.H, a.Lq, a.Lk, a.dk = 4, 1, 200, 200, 64
    assert L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()) != 0


@pytest.mark.parametrize("H,dk,kvdt", [(8, 64, 0), (8, 64, 1), (4, 16, 0)])
def test_attention_appends_new_key_to_cache(L, H, dk, kvdt):
    """k_new / v_new: the K / V of the new position become key Lk-1 — written to the cache row and attended to in the same
    call (row kernel), or appended by a helper launch first (every other kernel)."""
    rows, Lk, T = 21, 7, 18
    d = H * dk
    q, kn, vn = rnd(rows, d, seed=1), rnd(rows, d, seed=2), rnd(rows, d, seed=3)
    tdt = torch.bfloat16 if kvdt else torch.float32
    ck, cv = rnd(rows * T, d, seed=4).to(tdt), rnd(rows * T, d, seed=5).to(tdt)
    want_k, want_v = ck.clone().float(), cv.clone().float()
    slot = torch.arange(rows) * T + (Lk - 1)
    want_k[slot], want_v[slot] = kn, vn
    idx = (torch.arange(rows)[:, None] * T + torch.arange(Lk)[None, :])
    # the new key is used at full precision in this step; the cache holds it in its storage type
    kk = want_k[idx].view(rows, Lk, H, dk).transpose(1, 2); vv = want_v[idx].view(rows, Lk, H, dk).transpose(1, 2)
    ref = O.attention(q.view(rows, 1, H, dk).transpose(1, 2), kk, vv, None, None).transpose(1, 2).reshape(rows, d)
    a = L.AttnArgs()
    qd, knd, vnd, ckd, cvd = dev(q), dev(kn), dev(vn), dev(ck), dev(cv)
    o = torch.empty(rows, d, device="cuda")
    a.q, a.k, a.v, a.o = qd.data_ptr(), ckd.data_ptr(), cvd.data_ptr(), o.data_ptr()
    a.k_new, a.v_new, a.ld_new = knd.data_ptr(), vnd.data_ptr(), d
    a.ldq = a.ldk = a.ldv = a.ldo = d
    a.nkv, a.H, a.Lq, a.Lk, a.dk, a.kv_dtype, a.kv_group_stride = rows, H, 1, Lk, dk, kvdt, T
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    torch.testing.assert_close(o.cpu(), ref, rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(ckd.float().cpu(), want_k.to(tdt).float()); torch.testing.assert_close(cvd.float().cpu(), want_v.to(tdt).float())


@pytest.mark.parametrize("nkv,H,Lq,Lk,dk", [(3, 8, 85, 36, 64), (2, 8, 36, 36, 64), (4, 8, 17, 17, 64), (2, 2, 40, 20, 16)])
def test_attention_backward_in_two_parts(L, nkv, H, Lq, Lk, dk):
    """bwd_part 1 (dQ) then 2 (dK, dV) == bwd_part 0, whichever kernel serves the shape."""
    d = H * dk
    q, k, v, do = (dev(rnd(nkv * n, d, seed=s)) for n, s in ((Lq, 1), (Lk, 2), (Lk, 3), (Lq, 4)))
    o = torch.empty(nkv * Lq, d, device="cuda"); p = torch.empty(nkv, H, Lq, Lk, device="cuda")
    km = torch.ones(nkv, Lk); km[0, Lk - 3:] = 0
    kmd = dev(km)
    a = L.AttnArgs()
    a.q, a.k, a.v, a.o, a.p, a.kmask = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), p.data_ptr(), kmd.data_ptr()
    a.ldq = a.ldk = a.ldv = a.ldo = d
    a.nkv, a.H, a.Lq, a.Lk, a.dk = nkv, H, Lq, Lk, dk
    a.drop_p, a.drop_seed = 0.1, 5
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    outs = {}
    for parts in ((0,), (1, 2)):
        dq, dk_, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
        a.d_o, a.dq, a.d_k, a.dv = do.data_ptr(), dq.data_ptr(), dk_.data_ptr(), dv.data_ptr()
        a.lddo = a.lddq = a.lddk = a.lddv = d
        for part in parts:
            a.bwd_part = part
            L.check(L.lib().ortk_attention_bwd(C.byref(a), L.stream_ptr()), "attn_bwd")
        outs[parts] = (dq.clone(), dk_.clone(), dv.clone())
    for x, y in zip(outs[(0,)], outs[(1, 2)]):
        assert torch.isfinite(y).all() and torch.equal(x, y)


def test_attention_decode_row_kernel_with_ancestry_table(L):
    """Decode self-attention through the beam ancestry table (kv_index): row g attends to arbitrary cache rows."""
    rows, H, dk, Lk = 50, 8, 64, 11
    d = H * dk
    g = torch.Generator().manual_seed(3)
    q = rnd(rows, d, seed=1); cache_k = rnd(rows * 18, d, seed=2); cache_v = rnd(rows * 18, d, seed=3)
    idx = torch.randint(0, rows * 18, (rows, Lk), generator=g, dtype=torch.int32)
    kk = cache_k[idx.long()].view(rows, Lk, H, dk).transpose(1, 2); vv = cache_v[idx.long()].view(rows, Lk, H, dk).transpose(1, 2)
    ref = O.attention(q.view(rows, 1, H, dk).transpose(1, 2), kk, vv, None, None).transpose(1, 2).reshape(rows, d)
    a = L.AttnArgs()
    qd, kd, vd, idd = dev(q), dev(cache_k), dev(cache_v), dev(idx)
    for odt in (0, 1):
        o = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16 if odt else torch.float32)
        a.q, a.k, a.v, a.o, a.kv_index = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), idd.data_ptr()
        a.ldq = a.ldk = a.ldv = a.ldo = d
        a.nkv, a.H, a.Lq, a.Lk, a.dk, a.o_dtype = rows, H, 1, Lk, dk, odt
        L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
        torch.testing.assert_close(o.float().cpu(), ref, rtol=1e-2 if odt else 1e-4, atol=1e-2 if odt else 1e-5)


@pytest.mark.parametrize("Lq,Lk,use_idx", [(1, 13, True), (1, 30, False), (5, 36, False), (1, 36, False), (3, 20, False)])
def test_attention_bf16_kv_cache(L, Lq, Lk, use_idx):
    """kv_dtype = 1: K / V rows stored as bf16 (decode-time caches in mixed precision) == fp32 attention on the rounded K / V."""
    nkv, H, dk = 24, 8, 64
    d = H * dk
    g = torch.Generator().manual_seed(9)
    q = rnd(nkv * Lq, d, seed=1)
    pool = nkv * 40
    K16, V16 = rnd(pool, d, seed=2).bfloat16(), rnd(pool, d, seed=3).bfloat16()
    if use_idx:
        idx = torch.randint(0, pool, (nkv, Lk), generator=g, dtype=torch.int32)
    else:
        idx = (torch.arange(nkv)[:, None] * 40 + torch.arange(Lk)[None, :]).int()
    km = torch.ones(nkv, Lk); km[1, Lk // 2:] = 0
    kk = K16.float()[idx.long()].view(nkv, Lk, H, dk).transpose(1, 2); vv = V16.float()[idx.long()].view(nkv, Lk, H, dk).transpose(1, 2)
    ref = O.attention(q.view(nkv, Lq, H, dk).transpose(1, 2), kk, vv, km[:, None, None, :].bool(), None).transpose(1, 2).reshape(nkv * Lq, d)
    a = L.AttnArgs()
    qd, kd, vd, kmd = dev(q), dev(K16), dev(V16), dev(km)
    o = torch.empty(nkv * Lq, d, device="cuda")
    a.q, a.k, a.v, a.o, a.kmask = qd.data_ptr(), kd.data_ptr(), vd.data_ptr(), o.data_ptr(), kmd.data_ptr()
    a.ldq = a.ldk = a.ldv = a.ldo = d
    a.nkv, a.H, a.Lq, a.Lk, a.dk, a.kv_dtype = nkv, H, Lq, Lk, dk, 1
    if use_idx:
        idd = dev(idx); a.kv_index = idd.data_ptr()
    else:
        a.kv_group_stride = 40
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    torch.testing.assert_close(o.cpu(), ref, rtol=1e-4, atol=2e-5)
    a.Lq, a.Lk = 20, 20                                     # a shape no bf16-K/V kernel serves: refused, never misread
    assert L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()) == -1


@pytest.mark.parametrize("nkv,H,Lq,Lk,dk,prec", [(3, 2, 9, 11, 16, 0), (5, 8, 17, 17, 64, 0), (2, 8, 36, 36, 64, 0), (2, 8, 36, 36, 64, 1),
                                                 (2, 8, 85, 36, 64, 1)])
def test_attention_dropout_is_consistent_between_fwd_and_bwd(L, nkv, H, Lq, Lk, dk, prec):
    d = H * dk
    q, k, v, do = (dev(rnd(nkv * n, d, seed=s)) for n, s in ((Lq, 1), (Lk, 2), (Lk, 3), (Lq, 4)))
    a = L.AttnArgs()
    a.precision = prec
    o = torch.empty(nkv * Lq, d, device="cuda"); p = torch.empty(nkv, H, Lq, Lk, device="cuda")
    a.q, a.k, a.v, a.o, a.p = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), p.data_ptr()
    a.ldq = a.ldk = a.ldv = a.ldo = d
    a.nkv, a.H, a.Lq, a.Lk, a.dk = nkv, H, Lq, Lk, dk
    a.drop_p, a.drop_seed = 0.3, 77
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    # recover the mask from O = (P*m/keep) V by solving with the saved P: compare against a torch replay of the hash
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    a.d_o, a.dq, a.d_k, a.dv = do.data_ptr(), dq.data_ptr(), dk_.data_ptr(), dv.data_ptr()
    a.lddo = a.lddq = a.lddk = a.lddv = d
    L.check(L.lib().ortk_attention_bwd(C.byref(a), L.stream_ptr()), "attn_bwd")
    # finite-difference check of dV through the SAME dropout mask: O is linear in V
    eps = 0.5 if prec else 1e-2          # bf16 operands: the step must dwarf the rounding of v + eps
    v2 = v + eps * torch.ones_like(v)
    o2 = torch.empty_like(o); a.v, a.o = v2.data_ptr(), o2.data_ptr()
    L.check(L.lib().ortk_attention_fwd(C.byref(a), L.stream_ptr()), "attn_fwd")
    lhs = ((o2 - o) * do).sum().item() / eps
    assert abs(lhs - dv.sum().item()) < (3e-2 if prec else 1e-2) * max(1.0, abs(lhs))


def test_embed_xent_softmax_colsum(L):
    R, T, d, V = 6, 17, 64, 101
    g = torch.Generator().manual_seed(0)
    seq = torch.randint(0, V, (R, T + 1), generator=g); seq[:, 0] = 2; seq[2, 9:] = 0
    lut, pe = rnd(V, d, seed=1), O.positional_encoding(32, d)
    out = torch.empty(R * T, d, device="cuda"); km = torch.empty(R * T, device="cuda")
    L.check(L.lib().ortk_embed_fwd(L.ptr(dev(seq)), T + 1, L.ptr(dev(lut)), L.ptr(dev(pe)), L.ptr(out), L.ptr(km), R, T, 0, d, 0, 0.0, 0,
                                   L.stream_ptr()), "embed")
    ref = lut[seq[:, :T]] * math.sqrt(d) + pe[:T]
    torch.testing.assert_close(out.cpu().view(R, T, d), ref, rtol=1e-6, atol=1e-6)
    assert torch.equal(km.cpu().view(R, T), (seq[:, :T] != 0).float())
    dout = rnd(R * T, d, seed=2); dl = torch.zeros(V, d, device="cuda")
    L.check(L.lib().ortk_embed_bwd(L.ptr(dev(seq)), T + 1, L.ptr(dev(dout)), L.ptr(dl), R, T, d, 0.0, 0, L.stream_ptr()), "embed_bwd")
    refg = torch.zeros(V, d).index_add_(0, seq[:, :T].reshape(-1), dout * math.sqrt(d))
    torch.testing.assert_close(dl.cpu(), refg, rtol=1e-5, atol=1e-5)
    # fused CE
    ld = 104
    logits = rnd(R * T, ld, seed=3, scale=3.0)
    w = (seq[:, 1:] != 0).float() * (1 + rnd(R, 1, seed=4).abs())
    lg = logits[:, :V].clone().requires_grad_()
    lp = torch.log_softmax(lg, -1)
    norm = (seq[:, 1:] != 0).float().sum()
    ref_loss = -(lp.gather(1, seq[:, 1:].reshape(-1, 1)).squeeze(1) * w.reshape(-1)).sum() / norm
    ref_loss.backward()
    nd = torch.tensor([norm.item()], device="cuda"); wd = dev(w)
    seqd = dev(seq)  # keep alive
    loss = torch.full((1,), 7.0, device="cuda"); ld_dev = dev(logits)          # (*loss_dev is overwritten, not accumulated into)
    row_loss = torch.empty(L.lib().ortk_xent_scratch_floats(R * T), device="cuda")
    L.check(L.lib().ortk_xent_fwd_bwd(L.ptr(ld_dev), C.c_void_p(seqd.data_ptr() + 8), T + 1, T, L.ptr(wd), L.ptr(nd), L.ptr(loss),
                                      L.ptr(row_loss), R * T, V, ld, L.ptr(ld_dev), 0, ld, L.stream_ptr()), "xent")
    assert abs(loss.item() - ref_loss.item()) < 1e-5 * max(1, abs(ref_loss.item()))
    # the per-row terms, and the scalar = their sum in the documented fixed order (thread t adds rows t, t + 1024, ...; then a tree):
    # the same bits on every run
    ref_rows = -(lp.detach().gather(1, seq[:, 1:].reshape(-1, 1)).squeeze(1) * w.reshape(-1)) / norm
    torch.testing.assert_close(row_loss[:R * T].cpu(), ref_rows, rtol=1e-5, atol=1e-7)
    part = torch.zeros(1024); rl = row_loss[:R * T].cpu()
    for t in range(min(1024, R * T)):
        acc = torch.zeros((), dtype=torch.float32)
        for v in rl[t::1024]:
            acc = acc + v
        part[t] = acc
    k = 512
    while k >= 1:
        part[:k] = part[:k] + part[k:2 * k]
        k //= 2
    assert loss.item() == part[0].item(), (loss.item(), part[0].item())
    torch.testing.assert_close(ld_dev.cpu()[:, :V], lg.grad, rtol=1e-4, atol=1e-7)
    assert float(ld_dev[:, V:].abs().max()) == 0.0
    # log_softmax (+ temperature) and its backward
    x = dev(logits.clone())
    L.check(L.lib().ortk_log_softmax(L.ptr(x), R * T, V, ld, 0.5, L.stream_ptr()), "lsm")
    torch.testing.assert_close(x.cpu()[:, :V], torch.log_softmax(logits[:, :V] * 0.5, -1), rtol=1e-5, atol=1e-5)
    lpd = torch.log_softmax(logits[:, :V], -1)
    dlp = rnd(R * T, V, seed=5)
    lg2 = logits[:, :V].clone().requires_grad_(); torch.log_softmax(lg2, -1).backward(dlp)
    outg = torch.empty(R * T, ld, device="cuda")
    L.check(L.lib().ortk_log_softmax_bwd(L.ptr(dev(lpd)), L.ptr(dev(dlp)), V, L.ptr(outg), 0, ld, R * T, V, L.stream_ptr()), "lsm_bwd")
    torch.testing.assert_close(outg.cpu()[:, :V], lg2.grad, rtol=1e-4, atol=1e-5)
    # column sums
    X = rnd(1000, 77, seed=6); acc = dev(torch.ones(77))
    L.check(L.lib().ortk_colsum(L.ptr(dev(X)), 0, 77, L.ptr(acc), 1000, 77, L.stream_ptr()), "colsum")
    torch.testing.assert_close(acc.cpu(), 1 + X.sum(0), rtol=1e-4, atol=1e-4)
    # fixed-order scalar sum (the criterion's normaliser; the loss itself goes through the same reduction): one launch up to 65 536
    # elements, two stages beyond; bit-identical reruns, empty input = 0
    for n in (0, 1, 1023, 23040, 65536, 65537, 300001):
        xs = rnd(max(n, 1), seed=7 + n % 5)[:n].contiguous()
        ns = L.lib().ortk_sum_scratch_floats(n)
        assert ns == (0 if n <= 65536 else -(-n // 65536))
        scratch = torch.empty(max(ns, 1), device="cuda")
        outs = []
        for _ in range(3):
            o = torch.full((1,), 3.0, device="cuda")
            L.check(L.lib().ortk_sum(L.ptr(dev(xs)) if n else L.ptr(scratch), n, L.ptr(scratch) if ns else None, L.ptr(o), L.stream_ptr()), "sum")
            outs.append(o.item())
        assert outs[0] == outs[1] == outs[2]
        assert abs(outs[0] - xs.double().sum().item()) <= 1e-6 * max(1.0, xs.double().abs().sum().item())


def test_adam_clip_and_masks(L):
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.3)
    P = {"w": p.clone()}; st = {}
    base = torch.zeros(n + 5, device="cuda")     # odd length exercises the scalar tail
    pd = base[:n]; pd.copy_(p)
    md, vd = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for t in range(1, 4):
        O.adam_clip_step(P, {"w": g * t}, st, lr=0.01 * t, clip=0.1)
        L.check(L.lib().ortk_adam_clip(L.ptr(pd), L.ptr(dev(g * t)), L.ptr(md), L.ptr(vd), n, 0.01 * t, 0.9, 0.98, 1e-9, 0.1,
                                       1 - 0.9 ** t, 1 - 0.98 ** t, L.stream_ptr()), "adam")
    torch.testing.assert_close(pd.cpu(), P["w"], rtol=1e-5, atol=1e-6)
    # masks
    w, m, dwe = rnd(n, seed=3), rnd(n, seed=4, scale=2.0), rnd(n, seed=5)
    m[:10] = 0.0                                                # logit 0 -> sigmoid 0.5 -> round-half-even -> pruned
    we = torch.empty(n, device="cuda")
    L.check(L.lib().ortk_mask_apply(L.ptr(dev(w)), L.ptr(dev(m)), L.ptr(we), n, 0, 0, L.stream_ptr()), "mask")
    s = torch.round(torch.sigmoid(m))
    assert torch.equal(we.cpu(), s * w) and float(we[:10].abs().max()) == 0.0
    dw = torch.empty(n, device="cuda"); dm = torch.zeros(n, device="cuda"); coef = torch.tensor([0.37], device="cuda")
    L.check(L.lib().ortk_mask_bwd(L.ptr(dev(dwe)), L.ptr(dev(w)), L.ptr(dev(m)), L.ptr(dw), L.ptr(dm), n, 0, 0, L.ptr(coef),
                                  L.stream_ptr()), "mask_bwd")
    sg = torch.sigmoid(m)
    torch.testing.assert_close(dw.cpu(), dwe * s, rtol=0, atol=0)
    torch.testing.assert_close(dm.cpu(), (dwe * w + 0.37) * sg * (1 - sg), rtol=1e-5, atol=1e-7)
    cnt = torch.zeros(1, device="cuda")
    L.check(L.lib().ortk_mask_count(L.ptr(dev(m)), n, 0, L.ptr(cnt), L.stream_ptr()), "count")
    assert cnt.item() == s.sum().item()
    # bernoulli mode: keep-rate tracks sigmoid(m), and backward replays the same draw
    m2 = torch.full((n,), 0.8)
    L.check(L.lib().ortk_mask_apply(L.ptr(dev(torch.ones(n))), L.ptr(dev(m2)), L.ptr(we), n, 1, 99, L.stream_ptr()), "mask")
    assert abs(we.mean().item() - torch.sigmoid(torch.tensor(0.8)).item()) < 0.01
    L.check(L.lib().ortk_mask_bwd(L.ptr(dev(torch.ones(n))), L.ptr(dev(torch.ones(n))), L.ptr(dev(m2)), L.ptr(dw), None, n, 1, 99, None,
                                  L.stream_ptr()), "mask_bwd")
    assert torch.equal(dw, we)


@pytest.mark.parametrize("mode,train_masks,with_draws,with_active", [(1, True, False, True), (1, True, True, False), (0, True, False, False), (2, False, False, False)])
def test_masked_adam_step_equals_the_separate_launches(L, mode, train_masks, with_draws, with_active):
    """ortk_masked_adam_step (the element-wise tail of a masked training step in one pass over a RANGE of the arena) against the
    launches it replaces — ortk_mask_bwd[_draws] into a cleared dm, the frozen-scope multiply, ortk_adam_clip_zero on the weights,
    ortk_adam_clip on the mask logits — on the same inputs: weights, mask logits, all four moment arrays and the cleared gradient."""
    n, i0 = 100003, 4096          # a range that starts inside the arena: the hash draws are keyed by the arena position
    g = torch.Generator().manual_seed(mode + 10 * with_draws)
    mk = lambda scale=1.0: (torch.randn(i0 + n, generator=g) * scale)
    W, G, ML = mk(0.3), mk(0.05), (mk(2.0) if mode != 2 else (torch.rand(i0 + n, generator=g) < 0.3).float())
    MW, VW, MM, MV = mk(0.01), mk(0.01).abs(), mk(0.01), mk(0.01).abs()
    draws = torch.rand(i0 + n, generator=g) if with_draws else None
    active = (torch.rand(i0 + n, generator=g) < 0.7).float() if with_active else None
    coef = torch.tensor([0.37])
    hyper = dict(lr_w=3e-3, eps_w=1e-9, lr_m=100.0, eps_m=1e-2, b1=0.9, b2=0.98, clip=0.1, t=3)
    bc1, bc2 = 1 - hyper["b1"] ** hyper["t"], 1 - hyper["b2"] ** hyper["t"]
    seed = 0xABCDEF
    # reference: the separate launches over the WHOLE arena prefix [0, i0 + n) (the hash index is the arena position), range compared
    w1, g1, ml1, mw1, vw1, mm1, mv1 = (dev(t.clone()) for t in (W, G, ML, MW, VW, MM, MV))
    dm = torch.zeros(i0 + n, device="cuda")
    N = i0 + n
    if with_draws:
        L.check(L.lib().ortk_mask_bwd_draws(L.ptr(g1), L.ptr(w1), L.ptr(ml1), L.ptr(dev(draws)), L.ptr(g1), L.ptr(dm), N, L.ptr(dev(coef)), L.stream_ptr()), "bwd")
    else:
        L.check(L.lib().ortk_mask_bwd(L.ptr(g1), L.ptr(w1), L.ptr(ml1), L.ptr(g1), L.ptr(dm) if train_masks else None, N, mode, seed,
                                      L.ptr(dev(coef)), L.stream_ptr()), "bwd")
    if with_active:
        dm.mul_(dev(active))
    L.check(L.lib().ortk_adam_clip_zero(L.ptr(w1), L.ptr(g1), L.ptr(mw1), L.ptr(vw1), N, hyper["lr_w"], hyper["b1"], hyper["b2"], hyper["eps_w"],
                                        hyper["clip"], bc1, bc2, L.stream_ptr()), "adam w")
    if train_masks:
        L.check(L.lib().ortk_adam_clip(L.ptr(ml1), L.ptr(dm), L.ptr(mm1), L.ptr(mv1), N, hyper["lr_m"], hyper["b1"], hyper["b2"], hyper["eps_m"],
                                       hyper["clip"], bc1, bc2, L.stream_ptr()), "adam m")
    # one pass over the range [i0, i0 + n)
    w2, g2, ml2, mw2, vw2, mm2, mv2 = (dev(t.clone()) for t in (W, G, ML, MW, VW, MM, MV))
    keep = [dev(draws) if with_draws else None, dev(active) if with_active else None, dev(coef)]
    k = L.MaskedAdamArgs()
    sl = lambda t: t[i0:].data_ptr()
    k.w, k.g, k.mw, k.vw, k.ml = sl(w2), sl(g2), sl(mw2), sl(vw2), sl(ml2)
    if train_masks:
        k.mm, k.mv = sl(mm2), sl(mv2)
    if with_draws:
        k.draws = sl(keep[0])
    if with_active:
        k.active = sl(keep[1])
    k.extra_coef = keep[2].data_ptr()
    k.n, k.index0, k.mode, k.seed = n, i0, mode, seed
    k.lr_w, k.eps_w, k.lr_m, k.eps_m = hyper["lr_w"], hyper["eps_w"], hyper["lr_m"], hyper["eps_m"]
    k.beta1, k.beta2, k.clip, k.bc1, k.bc2 = hyper["b1"], hyper["b2"], hyper["clip"], bc1, bc2
    L.check(L.lib().ortk_masked_adam_step(C.byref(k), L.stream_ptr()), "ortk_masked_adam_step")
    torch.cuda.synchronize()
    for name, a_, b_ in (("w", w1, w2), ("mw", mw1, mw2), ("vw", vw1, vw2), ("ml", ml1, ml2), ("mm", mm1, mm2), ("mv", mv1, mv2)):
        orig = dev({"w": W, "mw": MW, "vw": VW, "ml": ML, "mm": MM, "mv": MV}[name])[i0:]
        d_ = (b_[i0:] - a_[i0:]).abs()
        # (the two kernels may contract a multiply-add differently: an ulp of the UPDATE — lr 100 on the mask logits — not of the result)
        tol = 2e-6 * (a_[i0:].abs() + (a_[i0:] - orig).abs()) + 1e-9
        assert bool((d_ <= tol).all()), (name, float(d_.max()), int((d_ > tol).sum()))
        assert torch.equal(b_[:i0], dev({"w": W, "mw": MW, "vw": VW, "ml": ML, "mm": MM, "mv": MV}[name])[:i0]), name      # nothing outside the range
    assert float(g2[i0:].abs().max()) == 0.0 and torch.equal(g2[:i0], dev(G)[:i0])
    if mode == 1 and not with_draws:
        assert float((w2[i0:] != dev(W)[i0:]).float().mean()) > 0.2          # (the sampled masks let a good part of the weights move)


def test_bf16_storage_paths(L):
    """Mixed-precision storage: bf16 A / B operands, bf16 C, bf16 LN / attention / dropout outputs vs fp32 math."""
    M, N, K = 300, 200, 192
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2)
    A16, B16 = dev(A).bfloat16(), dev(B).bfloat16()
    ref = A16.float().cpu() @ B16.float().cpu().t()
    sc = ref.abs().max().item()
    for (a_t, adt), (b_t, bdt) in [((A16, 1), (B16, 1)), ((dev(A), 0), (B16, 1)), ((A16, 1), (dev(B), 0))]:
        out = gemm(L, a_t, b_t, M, N, K, 0, 0, 1, a_dtype=adt, b_dtype=bdt)
        assert (out.cpu() - ref).abs().max().item() < 2e-2 * sc
    # k-major (wgrad / dgrad) layouts with bf16 operands
    At16, Bt16 = dev(A.t().contiguous()).bfloat16(), dev(B.t().contiguous()).bfloat16()
    out = gemm(L, At16, Bt16, M, N, K, 1, 1, 1, a_dtype=1, b_dtype=1)
    assert (out.cpu() - ref).abs().max().item() < 2e-2 * sc
    # fused column sums (bias gradient) in the wgrad layout: guarded shape, full-tile shape with split-K, fp32 A
    for (m_, n_, k_, adt_) in ((M, N, K, 1), (256, 128, 1024, 1), (256, 128, 1024, 0)):
        A_ = rnd(k_, m_, seed=21); B_ = rnd(k_, n_, seed=22)
        Ad = dev(A_).bfloat16() if adt_ else dev(A_); Bd = dev(B_).bfloat16()
        csum = torch.ones(m_, device="cuda"); Cacc = torch.zeros(m_, n_, device="cuda")
        gemm(L, Ad, Bd, m_, n_, k_, 1, 1, 1, a_dtype=adt_, b_dtype=1, C=Cacc, accumulate=1, splitk=4, colsum=csum)
        torch.testing.assert_close(csum.cpu(), 1 + Ad.float().cpu().sum(0), rtol=1e-3, atol=1e-2)
        refc = Ad.float().cpu().t() @ Bd.float().cpu()
        assert (Cacc.cpu() - refc).abs().max().item() < 2e-2 * refc.abs().max().item()
    out = gemm(L, A16, Bt16, M, N, K, 0, 1, 1, a_dtype=1, b_dtype=1)
    assert (out.cpu() - ref).abs().max().item() < 2e-2 * sc
    # bf16 C + bf16 gate
    gate = dev(rnd(M, N, seed=3)).bfloat16()
    C16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, C=C16, c_dtype=1, gate=gate, gate_dtype=1, gate_scale=2.0)
    refg = ref * (gate.float().cpu() > 0) * 2.0
    assert (C16.float().cpu() - refg).abs().max().item() < 3e-2 * sc
    # LayerNorm with bf16 output, dropout / gate / cast helpers
    rows, d = 77, 512
    x, a, b = rnd(rows, d, seed=4), 1 + 0.1 * rnd(d, seed=5), 0.1 * rnd(d, seed=6)
    y16 = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16); st = torch.empty(rows, 2, device="cuda")
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(dev(x)), L.ptr(dev(a)), L.ptr(dev(b)), L.ptr(y16), 1, L.ptr(st), rows, d, 1e-6, L.stream_ptr()), "ln16")
    torch.testing.assert_close(y16.float().cpu(), O.layer_norm(x, a, b), rtol=1e-2, atol=1e-2)
    xx = dev(rnd(1000, seed=7)); o16 = torch.empty(1000, device="cuda", dtype=torch.bfloat16)
    L.check(L.lib().ortk_dropout_apply(L.ptr(xx), L.ptr(o16), 1, 1000, 0.0, 0, L.stream_ptr()), "drop16")
    torch.testing.assert_close(o16.float(), xx.bfloat16().float())
    L.check(L.lib().ortk_cast_bf16(L.ptr(xx), L.ptr(o16), 1000, L.stream_ptr()), "cast")
    assert torch.equal(o16, xx.bfloat16())
    acc = torch.zeros(200, device="cuda")
    L.check(L.lib().ortk_colsum(L.ptr(C16), 1, N, L.ptr(acc), M, N, L.stream_ptr()), "colsum16")
    torch.testing.assert_close(acc.cpu(), C16.float().cpu().sum(0), rtol=1e-3, atol=1e-2)


def _ell_slot(k):
    return (k & ~7) | ((k + (k >> 3)) & 7)


def _ell_to_dense(plan, bi):
    """Decode block `bi` of a built plan back to a dense (N, K) fp32 matrix (inverse of the device builder)."""
    blk = plan.blocks[bi]
    N, K = blk["N"], blk["K"]
    c0, nch = plan._host[bi].chunk0, (N + 63) // 64
    cp, cl = plan.chunk_ptr.cpu().numpy(), plan.chunk_len.cpu().numpy()
    perm = plan.perm.cpu().numpy()
    st = plan.stream.cpu().numpy().view(np.uint32)
    inv = {_ell_slot(k): k for k in range(((K + 7) // 8) * 8)}
    W = np.zeros((N, K), np.float32)
    seen = set()
    for c in range(nch):
        ln, base = int(cl[c0 + c]), int(cp[c0 + c])
        assert ln % (8 if plan.format == 1 else 4) == 0 and base % 2 == 0
        for lane in range(64):
            col = int(perm[(c0 + c) * 64 + lane])
            if col < 0:
                continue
            assert col not in seen and c * 64 // 512 == col // 512          # every column once, inside its 512-column range
            seen.add(col)
            for j in range(ln):
                if plan.format == 1:      # ELL16 pairs: uint2 {off(even) | off(odd) << 16, bf16 w(even) | w(odd) << 16}
                    at = 2 * (base // 2 + (j // 2) * 64 + lane)
                    sh = 16 * (j & 1)
                    off = (int(st[at]) >> sh) & 0xFFFF
                    val = np.array([((int(st[at + 1]) >> sh) & 0xFFFF) << 16], np.uint32).view(np.float32)[0]
                else:
                    off = int(st[2 * (base + j * 64 + lane)])
                    val = st[2 * (base + j * 64 + lane) + 1:2 * (base + j * 64 + lane) + 2].view(np.float32)[0]
                if val != 0:
                    assert off % 16 == 0 and W[col, inv[off // 16]] == 0
                    W[col, inv[off // 16]] = val
    assert seen == set(range(N))
    return W


def _gu_to_dense(plan, bi):
    """Decode block `bi` of a built GU16 plan back to a dense (N, K) matrix, checking the invariants of the format: the 4
    address lanes of an entry agree, entries obey the residue rule that makes the LDS gathers conflict-free, every
    non-zero appears exactly once, dummies carry zero weights."""
    blk = plan.blocks[bi]
    N, K = blk["N"], blk["K"]
    slot0, G, nkc = plan._host[bi].chunk0, (N + 15) // 16, (K + 511) // 512
    wf = plan.stream.float().cpu().numpy().reshape(-1, 16, 64, 8)
    ko = plan.chunk_ptr.cpu().numpy().view(np.uint32).reshape(-1, 16, 64)
    ns = plan.chunk_len.cpu().numpy()
    W = np.zeros((((N + 15) // 16) * 16, K), np.float32)
    for g in range(G):
        for kc in range(nkc):
            slot = slot0 + g * nkc + kc
            S = int(ns[slot])
            assert 0 <= S <= 16
            for s in range(S):
                for lg in range(4):
                    for j in range(8):
                        ks = {(int(ko[slot, s, lg * 16 + 4 * (j & 3) + p]) >> (16 * (j >> 2))) & 0xFFFF for p in range(4)}
                        assert len(ks) == 1
                        krel = ks.pop()
                        e = 8 * lg + j
                        assert krel % 8 == ((e & 7) + 4 * ((e >> 3) & 1)) & 7 and krel < 512
                        col = wf[slot, s, lg * 16:(lg + 1) * 16, j]
                        if np.any(col != 0):
                            k = kc * 512 + krel
                            assert k < K and not np.any(W[16 * g:16 * g + 16, k] != 0)
                            W[16 * g:16 * g + 16, k] = col
    assert not np.any(W[N:] != 0)
    return W[:N]


@pytest.mark.parametrize("N,K,sp", [(130, 70, 0.8), (512, 512, 0.95), (1536, 512, 0.975), (512, 2048, 0.95), (5, 3, 0.5), (48, 1030, 0.9)])
def test_gu16_builder_round_trip(L, N, K, sp):
    """The GU16 builder (one wave per group of 16 outputs and chunk of 512 inputs) reproduces the bf16 weights exactly,
    from fp32 and bf16 sources, two blocks in one plan; format invariants checked by the decoder above."""
    from sparse_image_captioning_amd.sparse import SparsePlan
    g = torch.Generator().manual_seed(N + K)
    Ws = []
    for i in range(2):
        W = rnd(N, K, seed=1 + i, scale=0.2) * (torch.rand(N, K, generator=g) >= sp).float()
        W[N // 2] = 0.0
        Ws.append(W)
    dense = torch.cat([Ws[0].reshape(-1), torch.zeros(40), Ws[1].reshape(-1)])
    plan = SparsePlan([dict(offset=0, N=N, K=K, ld=K), dict(offset=N * K + 40, N=N, K=K, ld=K)], L.SP_GU16, "cuda")
    for src in (dev(dense), dev(dense).bfloat16()):
        plan.stream.fill_(7.0); plan.chunk_ptr.fill_(-1)
        plan.build(src)
        for bi in range(2):
            np.testing.assert_array_equal(_gu_to_dense(plan, bi), Ws[bi].bfloat16().float().numpy())
        assert plan.nnz == sum(int((w.bfloat16() != 0).sum()) for w in Ws)


@pytest.mark.parametrize("eb", [4, 8])
@pytest.mark.parametrize("N,K,sp", [(130, 70, 0.8), (512, 512, 0.95), (1536, 512, 0.95), (600, 2048, 0.97), (5, 3, 0.5)])
def test_ell_builder_round_trip(L, eb, N, K, sp):
    """The device builder (count / order / fill) reproduces the non-zeros of the dense block exactly: columns ordered by
    count inside each 512-column range, chunks padded to multiples of 4, two blocks in one plan, fp32 and bf16 sources."""
    from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
    fmt = L.SP_ELL16 if eb == 4 else L.SP_ELL32
    g = torch.Generator().manual_seed(N + K)
    Ws = []
    for i in range(2):
        W = rnd(N, K, seed=1 + i, scale=0.2) * (torch.rand(N, K, generator=g) >= sp).float()
        W[N // 2] = 0.0                               # an empty column list
        Ws.append(W)
    dense = torch.cat([Ws[0].reshape(-1), torch.zeros(37), Ws[1].reshape(-1)])
    off1 = N * K + 37
    blocks = [dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 1.3 * (1 - sp) + 0.05)),
              dict(offset=off1, N=N, K=K, ld=K, capacity=capacity_for(N, K, 1.3 * (1 - sp) + 0.05))]
    plan = SparsePlan(blocks, fmt, "cuda")
    for src in (dev(dense), dev(dense).bfloat16()):
        plan.stream.fill_(-1)
        plan.build(src)
        plan.check_overflow()
        for bi in range(2):
            want = Ws[bi] if (eb == 8 and src.dtype == torch.float32) else Ws[bi].bfloat16().float()
            got = _ell_to_dense(plan, bi)
            np.testing.assert_array_equal(got, want.numpy())
            cnt = plan.count.cpu().numpy()[bi * N:(bi + 1) * N]
            np.testing.assert_array_equal(cnt, (Ws[bi] != 0).sum(1).numpy())
    # a plan that is too small raises instead of silently dropping weights
    small = SparsePlan([dict(offset=0, N=N, K=K, ld=K, capacity=64)], fmt, "cuda")
    small.build(dev(dense))
    if int((Ws[0] != 0).sum()) > 64:
        with pytest.raises(Exception):
            small.check_overflow()


@pytest.mark.parametrize("xdt", [0, 1])
@pytest.mark.parametrize("fmt", [0, 1, 2])
@pytest.mark.parametrize("M,N,K,sp", [(300, 130, 70, 0.8), (5120, 512, 512, 0.95), (100, 512, 2048, 0.95), (33, 2048, 512, 0.9),
                                      (4000, 64, 600, 0.97), (1, 5, 3, 0.5), (900, 48, 1030, 0.9), (77, 1100, 512, 0.95),
                                      (9000, 512, 1536, 0.95), (40000, 96, 512, 0.9)])
def test_spmm_vs_dense(L, xdt, fmt, M, N, K, sp):
    """ortk_spmm (all three formats) == F.linear on the zero-filled weight (scripts/eval_model.py:64-88, masked_layer.py:134-135) with the
    GEMM's fused epilogue (bias / ReLU / row scale / dropout / gate / residual), fp32 and bf16 activations and outputs, ragged
    row counts, column counts that are not multiples of 4, 64 or 512."""
    from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
    eb = 8 if fmt == L.SP_ELL32 else 4                         # 8: fp32 values and activations; 4: bf16
    g = torch.Generator().manual_seed(M + N + K)
    W = rnd(N, K, seed=1, scale=0.2) * (torch.rand(N, K, generator=g) >= sp).float()
    W[N // 2] = 0.0
    X, bias, resid, rows = rnd(M, K, seed=2), rnd(N, seed=3), rnd(M, N, seed=4), torch.rand(M, generator=g)
    gate = rnd(M, N, seed=5)
    Xd = dev(X.bfloat16() if xdt else X)
    plan = SparsePlan([dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 1.3 * (1 - sp) + 0.05))], fmt, "cuda")
    plan.build(dev(W))
    plan.check_overflow()
    Wr = (W if eb == 8 else W.bfloat16().float()).double()
    Xr = (Xd.float().cpu() if eb == 8 else Xd.float().cpu().bfloat16().float()).double()
    for relu, use_res, ydt, use_rs, use_gate, drop in [(0, False, 0, False, False, 0.0), (1, True, 0, True, False, 0.0),
                                                       (1, False, 1, False, False, 0.0), (0, True, 0, False, True, 0.25)]:
        Y = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if ydt else torch.float32)
        a = L.SpmmArgs()
        a.X, a.Y, a.ldx, a.ldy, a.M, a.x_dtype, a.y_dtype = Xd.data_ptr(), Y.data_ptr(), K, N, M, xdt, ydt
        keep = [dev(bias), dev(resid), dev(rows), dev(gate)]
        a.bias, a.relu = keep[0].data_ptr(), relu
        if use_res:
            a.resid, a.ldr = keep[1].data_ptr(), N
        if use_rs:
            a.rowscale = keep[2].data_ptr()
        if use_gate:
            a.gate, a.ldg, a.gate_dtype, a.gate_scale = keep[3].data_ptr(), N, 0, 1.5
        a.drop_p, a.drop_seed = drop, 1234
        plan.spmm(0, a)
        ref = Xr @ Wr.t() + bias.double()
        if relu:
            ref = ref.clamp_min(0)
        if use_rs:
            ref = ref * rows.double()[:, None]
        got = Y.float().cpu().double()
        assert torch.isfinite(got).all()
        if drop > 0:          # the keep pattern is the GEMM's (ortk_keep): compare through the same kernel family
            Yg = gemm(L, dev(Xd.float()), dev(W), M, N, K, bias=dev(bias + 100.0), drop_p=drop, drop_seed=1234)
            kept = (Yg != 0).cpu()
            ref = torch.where(kept, ref / (1 - drop), torch.zeros_like(ref))
            assert 0.6 < kept.float().mean().item() < 0.9 or M * N < 100
        if use_gate:
            ref = torch.where(gate > 0, ref * 1.5, torch.zeros_like(ref))
        if use_res:
            ref = ref + resid.double()
        tol = 2e-2 if ydt else 2e-5
        err = (got - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err < tol, (relu, use_res, ydt, use_rs, use_gate, drop, err)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (1280, 512, 2048), (384, 256, 192), (2048, 2048, 192), (4096, 1024, 64), (4096, 2560, 192), (3328, 3072, 64),
                                   (4480, 2560, 512), (5376, 512, 128),
                                   # ragged row counts (any batch size): LDS-DMA kernels with a clamped, bounds-checked last row tile
                                   (1000, 512, 512), (37, 1536, 512), (4500, 2560, 192), (10795, 512, 512), (21590, 512, 128), (635, 2048, 512)])
def test_gemm_full_tile_bf16_pipeline(L, M, N, K):
    """Full-tile bf16 x bf16 shapes take the LDS-DMA pipelined kernel (swizzled images, 4-stage ring): all three operand
    layouts, fused epilogue, split-K accumulation and fused column sums — against fp32 matmuls of the same bf16 values
    (fp32 accumulation: tolerance 1e-3 of the output scale, an order of magnitude below bf16 rounding)."""
    A, B = rnd(M, K, seed=11), rnd(N, K, seed=12)
    A16, B16 = dev(A).bfloat16(), dev(B).bfloat16()
    ref = A16.float().cpu().double() @ B16.float().cpu().double().t()
    sc = ref.abs().max().item()
    out = gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1)                         # X W^T
    assert (out.cpu().double() - ref).abs().max().item() < 1e-3 * sc
    Bt16 = B16.t().contiguous()
    out = gemm(L, A16, Bt16, M, N, K, 0, 1, 1, a_dtype=1, b_dtype=1)                        # dY W
    assert (out.cpu().double() - ref).abs().max().item() < 1e-3 * sc
    At16 = A16.t().contiguous()
    out = gemm(L, At16, Bt16, M, N, K, 1, 1, 1, a_dtype=1, b_dtype=1)                       # dY^T X
    assert (out.cpu().double() - ref).abs().max().item() < 1e-3 * sc
    # epilogue: bias + relu + residual, bf16 output
    bias, resid = rnd(N, seed=13), rnd(M, N, seed=14)
    C16 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, C=C16, c_dtype=1, bias=dev(bias), relu=1, resid=dev(resid))
    refe = (ref + bias.double()).clamp_min(0) + resid.double()
    assert (C16.float().cpu().double() - refe).abs().max().item() < 1e-2 * max(sc, 1.0)
    # dropout epilogue (grouped hash): reproducible, right keep rate, kept entries scaled
    d1 = gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, drop_p=0.2, drop_seed=99)
    d2 = gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, drop_p=0.2, drop_seed=99)
    assert torch.equal(d1, d2)
    keptm = d1.cpu() != 0
    assert abs(keptm.float().mean().item() - 0.8) < 0.02
    assert ((d1.cpu().double() - ref / 0.8).abs()[keptm]).max().item() < 2e-3 * sc
    # split-K accumulation on top of existing content + fused column sums of A
    for sk in (1, 2, 3):
        Cacc = torch.ones(M, N, device="cuda"); csum = torch.full((M,), 2.0, device="cuda")
        gemm(L, At16, Bt16, M, N, K, 1, 1, 1, a_dtype=1, b_dtype=1, C=Cacc, accumulate=1, splitk=sk, colsum=csum)
        assert (Cacc.cpu().double() - 1 - ref).abs().max().item() < 1e-3 * sc
        torch.testing.assert_close(csum.cpu(), 2 + At16.float().cpu().sum(0), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize("M,N,K", [(16640, 512, 128), (4500, 2560, 192), (21590, 512, 64),      # 256 x 256 tiles (whole, ragged last row tile)
                                   (1000, 512, 512), (9216, 1536, 64), (37, 2048, 128),            # 128 x 128 tiles
                                   (16640, 640, 128), (16500, 896, 64)])                           # 256 n + 128 columns: big tiles + a 128-column remainder launch
def test_gemm_lean_epilogue_equals_general(L, M, N, K):
    """The lean epilogue of the forward-layout LDS-DMA kernels (options compiled out, dropout / gate as kernel instances, permuted column
    order with 16-byte bf16 stores; ortk_tuning.gemm_epilogue = 0) gives BIT-identical results to the general epilogue it replaces
    (gemm_epilogue = 1: same arithmetic in the same order), for every option the executor's forward-layout products use — bias, ReLU,
    residual, dropout (with a row map, stride and offset), gate in both element types, both result types — and both agree with torch."""
    A16, B16 = dev(rnd(M, K, seed=21).bfloat16()), dev(rnd(N, K, seed=22).bfloat16())
    bias, resid, gate = dev(rnd(N, seed=23)), dev(rnd(M, N, seed=24)), rnd(M, N, seed=25)
    g = torch.Generator().manual_seed(5)
    rows = dev(torch.randperm(M, generator=g).int())
    combos = [dict(), dict(bias=bias, relu=1, c_dtype=1), dict(bias=bias, resid=resid), dict(resid=resid, c_dtype=1, relu=1),
              dict(bias=bias, relu=1, drop_p=0.1, drop_seed=77, c_dtype=1), dict(drop_p=0.3, drop_seed=5, drop_rows=rows, drop_row_stride=3, drop_row_off=1, resid=resid),
              dict(gate=dev(gate), gate_scale=1.25), dict(gate=dev(gate.bfloat16()), gate_dtype=1, gate_scale=1.0 / 0.9, c_dtype=1, bias=bias),
              dict(gate=dev(gate), gate_scale=2.0, drop_p=0.2, drop_seed=9, resid=resid, bias=bias, relu=1)]
    ref0 = (A16.float() @ B16.float().t())
    try:
        for kw in combos:
            out = []
            for general in (0, 1):
                L.set_tuning(gemm_epilogue=general)
                Cx = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16 if kw.get("c_dtype") else torch.float32)
                out.append(gemm(L, A16, B16, M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, C=Cx, **kw).clone())
            assert torch.equal(out[0], out[1]), sorted(kw)
            if "drop_p" not in kw:
                ref = ref0 + (bias if "bias" in kw else 0)
                if kw.get("relu"): ref = ref.clamp_min(0)
                if "gate" in kw: ref = torch.where(dev(gate) > 0, ref * kw["gate_scale"], torch.zeros_like(ref))
                if "resid" in kw: ref = ref + resid
                err = (out[0].float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
                assert err < (1e-2 if kw.get("c_dtype") else 1e-4), (sorted(kw), err)
            else:
                keep = (out[0].float() != (resid if "resid" in kw else 0)).float().mean().item()
                if "gate" not in kw and not kw.get("relu"): assert abs(keep - (1 - kw["drop_p"])) < 0.02, keep
    finally:
        L.set_tuning(gemm_epilogue=0)


def _ln_gemm_inputs(M, K, seed):
    A = rnd(M, K, seed=seed).bfloat16()
    W = rnd(512, K, seed=seed + 1, scale=K ** -0.5).bfloat16()
    return A, W


@pytest.mark.parametrize("M,K,p", [(16640, 512, 0.1), (1000, 2048, 0.0), (129, 64, 0.1), (77, 512, 0.0)])
def test_gemm_with_fused_layernorm_forward(L, M, K, p):
    """ortk_gemm(ln_mode = 1) — projection + residual + the next sublayer's LayerNorm in one launch (the 128 x 512 row-panel
    kernel) — against the same call without ln_mode followed by ortk_layernorm_fwd.  The GEMM result is the same MFMA
    arithmetic in a different accumulation order (1e-5 of the scale); the LayerNorm is fp32 on that result."""
    A, W = _ln_gemm_inputs(M, K, 20)
    bias, res, ga, gb = rnd(512, seed=22), rnd(M, 512, seed=23), 1 + 0.1 * rnd(512, seed=24), 0.1 * rnd(512, seed=25)
    kw = dict(a_dtype=1, b_dtype=1, bias=dev(bias), resid=dev(res), drop_p=p, drop_seed=77)
    Ad, Wd = dev(A), dev(W)
    c_ref = gemm(L, Ad, Wd, M, 512, K, 0, 0, 1, **kw)
    y_ref = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16); st_ref = torch.empty(M, 2, device="cuda")
    gad, gbd = dev(ga), dev(gb)
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(c_ref), L.ptr(gad), L.ptr(gbd), L.ptr(y_ref), 1, L.ptr(st_ref), M, 512, 1e-6, L.stream_ptr()), "ln")
    y = torch.full((M, 512), float("nan"), device="cuda", dtype=torch.bfloat16); st = torch.full((M, 2), float("nan"), device="cuda")
    c = gemm(L, Ad, Wd, M, 512, K, 0, 0, 1, ln_mode=1, ln_y_dtype=1, ln_a=gad, ln_b=gbd, ln_y=y, ln_stats=st, ln_eps=1e-6, **kw)
    torch.cuda.synchronize()
    scale = c_ref.abs().max().item()
    assert (c - c_ref).abs().max().item() <= 1e-5 * scale
    assert (c == 0).float().mean().item() == pytest.approx((c_ref == 0).float().mean().item(), abs=1e-6)     # same dropout draws
    torch.testing.assert_close(st, st_ref, rtol=1e-5, atol=1e-5)
    assert (y.float() - y_ref.float()).abs().max().item() <= 2 ** -7 * y_ref.float().abs().max().item()      # one bf16 ulp at the top


@pytest.mark.parametrize("M,K,p,with_res", [(16640, 512, 0.1, True), (1000, 2048, 0.0, True), (129, 64, 0.1, False), (77, 1536, 0.0, True)])
def test_gemm_with_fused_layernorm_backward(L, M, K, p, with_res):
    """ortk_gemm(ln_mode = 2) — data-gradient product + LayerNorm backward + residual-gradient add + the masked bf16 copy —
    against ortk_gemm followed by ortk_layernorm_bwd_drop."""
    A, W = _ln_gemm_inputs(M, K, 30)
    x, ga, gb, dres = rnd(M, 512, seed=32), 1 + 0.1 * rnd(512, seed=33), 0.1 * rnd(512, seed=34), rnd(M, 512, seed=35)
    xd, gad, gbd, rd = dev(x), dev(ga), dev(gb), dev(dres)
    y = torch.empty(M, 512, device="cuda"); st = torch.empty(M, 2, device="cuda")
    L.check(L.lib().ortk_layernorm_fwd(L.ptr(xd), L.ptr(gad), L.ptr(gbd), L.ptr(y), 0, L.ptr(st), M, 512, 1e-6, L.stream_ptr()), "ln")
    Ad, Wd = dev(A), dev(W)
    dy = gemm(L, Ad, Wd, M, 512, K, 0, 0, 1, a_dtype=1, b_dtype=1)
    dx_ref = torch.empty(M, 512, device="cuda"); dz_ref = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16)
    da_ref = torch.zeros(512, device="cuda"); db_ref = torch.zeros(512, device="cuda")
    L.check(L.lib().ortk_layernorm_bwd_drop(L.ptr(dy), L.ptr(xd), L.ptr(gad), L.ptr(st), L.ptr(rd) if with_res else None, L.ptr(dx_ref),
                                            L.ptr(da_ref), L.ptr(db_ref), M, 512, 1e-6, L.ptr(dz_ref), 1, p, 99, L.stream_ptr()), "lnb")
    dz = torch.full((M, 512), float("nan"), device="cuda", dtype=torch.bfloat16)
    da = torch.zeros(512, device="cuda"); db = torch.zeros(512, device="cuda")
    kw = dict(ln_dres=rd) if with_res else {}
    dx = gemm(L, Ad, Wd, M, 512, K, 0, 0, 1, a_dtype=1, b_dtype=1, ln_mode=2, ln_y_dtype=1, ln_a=gad, ln_y=dz, ln_stats=st, ln_eps=1e-6,
              ln_x=xd, ln_da=da, ln_db=db, drop_p=p, drop_seed=99, **kw)
    torch.cuda.synchronize()
    s = dx_ref.abs().max().item()
    assert (dx - dx_ref).abs().max().item() <= 2e-5 * s
    assert (dz.float() - dz_ref.float()).abs().max().item() <= 2 ** -7 * dz_ref.float().abs().max().item()
    assert ((dz == 0) == (dz_ref == 0)).float().mean().item() > 1 - 1e-5
    torch.testing.assert_close(da, da_ref, rtol=1e-4, atol=1e-4 * da_ref.abs().max().item())
    torch.testing.assert_close(db, db_ref, rtol=1e-4, atol=1e-4 * db_ref.abs().max().item())


def test_gemm_fused_layernorm_fallback_shapes(L):
    """Shapes the row-panel kernel does not take (N != 512, fp32 operands) give the same results through the separate kernels."""
    M, N, K = 50, 96, 80
    A, W, ga, gb = rnd(M, K, seed=40), rnd(N, K, seed=41), 1 + 0.1 * rnd(N, seed=42), 0.1 * rnd(N, seed=43)
    y = torch.empty(M, N, device="cuda"); st = torch.empty(M, 2, device="cuda")
    c = gemm(L, dev(A), dev(W), M, N, K, ln_mode=1, ln_y_dtype=0, ln_a=dev(ga), ln_b=dev(gb), ln_y=y, ln_stats=st, ln_eps=1e-6)
    ref = A @ W.t()
    torch.testing.assert_close(c.cpu(), ref, rtol=2e-5, atol=2e-4)
    torch.testing.assert_close(y.cpu(), O.layer_norm(ref, ga, gb), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("M,V", [(300, 1000), (129, 10000)])
def test_gemm_tile_softmax_partials(L, M, V):
    """ortk_gemm_args.tile_stats: {max, sum exp(. - max)} per block of 64 output columns (bias included, columns >= stat_ncols
    left out) — what the beam step reads instead of the logits.  Against torch on the GEMM's own fp32 output."""
    N, K = ((V + 127) // 128) * 128, 512
    A = rnd(M, K, seed=50).bfloat16()
    W = torch.zeros(N, K); W[:V] = rnd(V, K, seed=51, scale=K ** -0.5) * 3
    bias = torch.zeros(N); bias[:V] = rnd(V, seed=52)
    nblk = N // 64
    stats = torch.full((M, nblk, 2), float("nan"), device="cuda")
    c = gemm(L, dev(A), dev(W.bfloat16()), M, N, K, 0, 0, 1, a_dtype=1, b_dtype=1, bias=dev(bias), tile_stats=stats, stat_ncols=V)
    torch.cuda.synchronize()
    x = c.clone(); x[:, V:] = float("-inf")
    xb = x.view(M, nblk, 64)
    mx = xb.amax(2)
    sm = torch.where(mx.isinf(), torch.zeros_like(mx), torch.exp(xb - mx.clamp(min=-1e30)[..., None]).sum(2))
    assert torch.equal(stats[..., 0], mx)
    torch.testing.assert_close(stats[..., 1], sm, rtol=2e-6, atol=1e-6)
    # the row's log-sum-exp from the partials
    lse = torch.logsumexp(x[:, :V], 1)
    m_row = stats[..., 0].amax(1)
    lse2 = m_row + torch.log((stats[..., 1] * torch.exp(stats[..., 0] - m_row[:, None])).sum(1))
    torch.testing.assert_close(lse2, lse, rtol=1e-6, atol=2e-6)


@pytest.mark.parametrize("M,V,K_greedy,temp,constraint", [(300, 1000, 0, 1.0, 0), (258, 10001, 6, 1.0, 1), (129, 5000, 3, 0.7, 1)])
def test_gemm_gumbel_max_candidates(L, M, V, K_greedy, temp, constraint):
    """ortk_gemm_args.tile_samp: per row and block of 64 logits the Gumbel-max candidate {key, column, logit} the sampling decode
    combines instead of reading the logit rows (caption_model.py:56-111's multinomial draw as an arg-max over logit / T + Gumbel noise;
    the noise is oracle/ort_oracle.py: gumbel_from_hash).  Against the same launch's own fp32 logits: every block's candidate IS the
    arg-max of key(m, v) over the block's admissible columns (exact: same floats, same total order), greedy rows take the plain
    logits, the previous token is excluded, and the arg-max over the blocks equals the row's arg-max."""
    N, Kd, t, seed, off = ((V + 127) // 128) * 128, 512, 3, 77, 1000
    A = rnd(M, Kd, seed=60).bfloat16()
    W = torch.zeros(N, Kd); W[:V] = rnd(V, Kd, seed=61, scale=Kd ** -0.5) * 3
    bias = torch.zeros(N); bias[:V] = rnd(V, seed=62)
    nblk, T = N // 64, 8
    seqs = torch.randint(0, V, (M, T), generator=torch.Generator().manual_seed(5))
    stats = torch.full((M, nblk, 2), float("nan"), device="cuda"); cand = torch.full((M, nblk, 4), float("nan"), device="cuda")
    kw = dict(samp_seq=dev(seqs)) if constraint else {}
    c = gemm(L, dev(A), dev(W.bfloat16()), M, N, Kd, 0, 0, 1, a_dtype=1, b_dtype=1, bias=dev(bias), tile_stats=stats, stat_ncols=V, tile_samp=cand,
             samp_seed=seed, samp_row_offset=off, samp_L=T, samp_t=t, samp_greedy_stride=K_greedy, samp_sample=1, samp_fast=0,
             samp_inv_temperature=1.0 / temp, **kw)
    torch.cuda.synchronize()
    z = c.cpu()[:, :V]
    rows = torch.arange(M)
    greedy = (rows % K_greedy == 0) if K_greedy else torch.zeros(M, dtype=torch.bool)
    grow = rows + off
    hrow = (grow - grow // K_greedy - 1) if K_greedy else grow
    g = torch.stack([O.gumbel_from_hash(seed, t, 1, V, int(h))[0] for h in hrow])
    key = torch.where(greedy[:, None], z, z * torch.tensor(1.0 / temp, dtype=torch.float32) + g)
    if constraint:
        key[rows, seqs[:, t - 1]] = float("-inf")
    keyp = torch.full((M, N), float("-inf")); keyp[:, :V] = key
    kb = keyp.view(M, nblk, 64)
    ref_key, ref_idx = kb.max(2)
    ref_idx = ref_idx + 64 * torch.arange(nblk)[None, :]
    got = cand.cpu()
    gi = got[..., 1].contiguous().view(torch.int32)
    live = ~ref_key.isinf()
    # the Gumbel noise comes from libm logf on the device against torch.log on the host: keys agree to 1e-6, ties aside the columns agree
    assert (got[..., 0][live] - ref_key[live]).abs().max().item() < 2e-5
    agree = (gi[live] == ref_idx[live].int()).float().mean().item()
    assert agree > 0.999, agree
    same = live & (gi == ref_idx.int())
    assert torch.equal(got[..., 2][same], z[rows[:, None].expand(-1, nblk)[same], ref_idx[same]])
    # the arg-max over the blocks is the row's arg-max
    best_blk = got[..., 0].argmax(1)
    tok = gi[rows, best_blk]
    assert (tok.long() == key.argmax(1)).float().mean().item() > 0.99


# ------------------------------------------------------------------------------------------------ rows-stationary chains (round 4)
def _keep_mask(L, seed, n, p):
    """keep / (1 - p) per element of a dropout site, as the kernels draw it (counter hash, element index = row * N + col)."""
    ones, out = torch.ones(n, device="cuda"), torch.empty(n, device="cuda")
    L.check(L.lib().ortk_dropout_apply(L.ptr(ones), L.ptr(out), 0, n, p, seed, L.stream_ptr()), "ortk_dropout_apply")
    return out


def _ln_ref(x, g, b, eps=1e-6):
    mean = x.mean(-1, keepdim=True)
    sd = x.std(-1, keepdim=True)                      # unbiased (transformer.py:338-341)
    return g * (x - mean) / (sd + eps) + b, mean.squeeze(-1), sd.squeeze(-1)


@pytest.mark.parametrize("M,p,parts,pf,keyed", [(1000, 0.1, "R1SFL2S", 1, 0), (16640, 0.1, "R1SFL2S", 1, 0), (9216, 0.0, "R1SFL2S", 1, 0), (777, 0.1, "1S", 1, 0),
                                                (2000, 0.1, "R1S", 0, 0), (5120, 0.0, "R1F2", 1, 0),
                                                # the 76-row (four-wave) form of the kernel: chosen where it saves a round of workgroups
                                                (16640, 0.1, "R1SFL2S", 0, 0), (12300, 0.1, "R1SFL2S", 0, 0), (19456, 0.0, "R1F2", 0, 0), (13000, 0.1, "R1S", 0, 0),
                                                (36864, 0.1, "1S", 0, 0),
                                                # ortk_chain_args.drop_rows: row m draws the dropout of row drop_rows[m] (both kernel forms)
                                                (1000, 0.1, "R1SFL2S", 1, 1), (14000, 0.1, "R1SFL2S", 0, 1)])
def test_row_chain_vs_separate_ops(L, M, p, parts, pf, keyed):
    """ortk_row_chain (csrc/ortk_chain.hip: the row-wise operators between two attention calls in ONE rows-stationary launch) against
    the same chain in torch fp32 on the bf16-rounded operands, with the kernels' own dropout masks: residual streams within 2e-3
    (fp32 accumulation order), bf16 outputs within one bf16 step, LayerNorm statistics within 1e-4.  Chains: the decoder's
    [Wo -> LN -> W1 .. W2 -> LN -> Wqkv] at 1 000 / 16 640 rows (33-row blocks, two rounds) and 9 216 rows (36-row blocks), a bare
    [LN -> W] prefix, [Wo -> LN -> Wcq], and a chain that ends in a LayerNorm (the stack's last layer); the same shapes through the
    76-row form of the kernel (65-, 49-, 76- and 51-row blocks: whole and partial row tiles; 36 864 rows stay on the 48-row form)."""
    d, NC = 512, 4
    ff = NC * d
    hasR, n1, hasF = "R" in parts, (1 if parts in ("R1S",) else 3 if "1S" in parts else 0), "F" in parts
    has2, n2 = ("2" in parts), (3 if parts.endswith("2S") else 0)
    g = torch.Generator().manual_seed(M)
    nblk = (1 if hasR else 0) + n1 + (2 if hasF else 0) * 1 + n2
    # one bf16 "arena": [Wr (512,512)] [S1 (n1*512, 512)] [W1 (ff,512)] [W2 (512,ff)] [S2 (n2*512,512)]
    shapes = []
    if hasR: shapes.append(("r", d, d))
    if n1: shapes.append(("s1", n1 * d, d))
    if hasF: shapes += [("w1", ff, d), ("w2", d, ff)]
    if n2: shapes.append(("s2", n2 * d, d))
    W, off, cur = {}, {}, 0
    for name, N, K in shapes:
        W[name] = (torch.randn(N, K, generator=g) * (0.5 / math.sqrt(K))).bfloat16()
        off[name] = cur
        cur += N * K
    arena = dev(torch.cat([W[n].reshape(-1) for n, _, _ in shapes]))
    units = []
    if hasR: units.append((off["r"], d))
    units += [(off["s1"] + i * d * d, d) for i in range(n1)]
    if hasF:
        for c in range(NC):
            units += [(off["w1"] + c * d * d, d), (off["w2"] + c * d, ff)]
    units += [(off["s2"] + i * d * d, d) for i in range(n2)]
    ut = dev(torch.tensor(units, dtype=torch.int64))
    x = dev(rnd(M, d, seed=1))
    a_in = dev(rnd(M, d, seed=2).bfloat16())
    vec = lambda n, s, sc=0.1: dev(rnd(n, seed=s, scale=sc))
    br, g1, b1, bs1, bh, bo, g2, b2, bs2 = vec(d, 3), dev(1 + rnd(d, seed=4, scale=0.1)), vec(d, 5), vec(3 * d, 6), vec(ff, 7), vec(d, 8), dev(1 + rnd(d, seed=9, scale=0.1)), vec(d, 10), vec(3 * d, 11)
    x_mid, x_out = torch.full((M, d), float("nan"), device="cuda"), torch.full((M, d), float("nan"), device="cuda")
    y1, y2 = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16), torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
    st1, st2 = torch.zeros(M, 2, device="cuda"), torch.zeros(M, 2, device="cuda")
    out1 = torch.zeros(M, max(n1, 1) * d, device="cuda", dtype=torch.bfloat16)
    out2 = torch.zeros(M, max(n2, 1) * d, device="cuda", dtype=torch.bfloat16)
    h = torch.zeros(M, ff, device="cuda", dtype=torch.bfloat16)
    a = L.ChainArgs()
    a.w16, a.units_dev, a.n_units = arena.data_ptr(), ut.data_ptr(), len(units)
    nb = L.lib().ortk_chain_packed_bytes(len(units))
    packed = torch.empty(nb, dtype=torch.uint8, device="cuda")
    a.packed, a.packed_bytes = packed.data_ptr(), nb
    a.M, a.x_in = M, x.data_ptr()
    sr, sh, so = 1111, 2222, 3333
    if hasR:
        a.a_in, a.bias_r, a.x_mid, a.seed_r = a_in.data_ptr(), br.data_ptr(), x_mid.data_ptr(), sr
    a.g1, a.b1, a.y1, a.st1 = g1.data_ptr(), b1.data_ptr(), y1.data_ptr(), st1.data_ptr()
    a.n1 = n1
    if n1:
        a.bias_s1, a.out1, a.ld1 = bs1.data_ptr(), out1.data_ptr(), out1.stride(0)
    if hasF:
        a.NC, a.bias_h, a.bias_o, a.h, a.x_out, a.seed_h, a.seed_o = NC, bh.data_ptr(), bo.data_ptr(), h.data_ptr(), x_out.data_ptr(), sh, so
    if has2:
        a.g2, a.b2, a.y2, a.st2 = g2.data_ptr(), b2.data_ptr(), y2.data_ptr(), st2.data_ptr()
    a.n2 = n2
    if n2:
        a.bias_s2, a.out2, a.ld2 = bs2.data_ptr(), out2.data_ptr(), out2.stride(0)
    a.drop_p, a.eps = p, 1e-6
    # keyed: rows draw as OTHER rows of a taller (Mk, .) tensor — what the valid-position decoder layout does with ortk_batch.row_pos
    Mk = M + 500 if keyed else M
    key_rows = torch.randperm(Mk, generator=g)[:M].sort().values if keyed else torch.arange(M)
    key_dev = dev(key_rows.to(torch.int32))
    if keyed:
        a.drop_rows = key_dev.data_ptr()
    key_idx = key_rows.cuda()
    prog = torch.zeros(16, dtype=torch.int32, device="cuda")
    if pf:                               # with the L2 prefetcher workgroups (always the 48-row form)
        a.progress = prog.data_ptr()
    # (the 76-row form is the default where ONE round of its blocks replaces two of the 48-row form: 12 289 .. 19 456 rows; the cases with
    #  the prefetcher workgroups and the row counts outside that range run the 48-row form)
    L.check(L.lib().ortk_row_chain(C.byref(a), L.stream_ptr()), "ortk_row_chain")
    torch.cuda.synchronize()
    # ---- reference (fp32 on the GPU, operands rounded to bf16 where the kernel rounds them)
    f = lambda t: t.float().cuda()
    xr = x.clone()
    one_bf16 = lambda ref: 2.0 ** -7 * ref.abs().clamp(min=1.0)          # one step of an 8-bit mantissa (+ slack for a rounding boundary)
    if hasR:
        t = a_in.float() @ f(W["r"]).t() + br
        if p > 0: t = t * _keep_mask(L, sr, Mk * d, p).view(Mk, d)[key_idx]
        xr = xr + t
        assert (x_mid - xr).abs().max().item() < 2e-3
    yr, mean, sd = _ln_ref(xr, g1, b1)
    assert (st1[:, 0] - mean).abs().max().item() < 1e-4 and (st1[:, 1] - sd).abs().max().item() < 1e-4
    assert ((y1.float() - yr).abs() <= one_bf16(yr)).all()
    yb = y1.float()                                     # the kernel's own rounded rows feed the products
    if n1:
        ref = yb @ f(W["s1"]).t() + bs1[:n1 * d]
        assert ((out1.float() - ref).abs() <= one_bf16(ref) + 2e-3).all()
    if hasF:
        hr = torch.relu(yb @ f(W["w1"]).t() + bh)
        if p > 0: hr = hr * _keep_mask(L, sh, Mk * ff, p).view(Mk, ff)[key_idx]
        assert ((h.float() - hr).abs() <= one_bf16(hr) + 2e-3).all()
        t = h.float() @ f(W["w2"]).t() + bo
        if p > 0: t = t * _keep_mask(L, so, Mk * d, p).view(Mk, d)[key_idx]
        xr = xr + t
        assert (x_out - xr).abs().max().item() < 4e-3
    if has2:
        yr, mean, sd = _ln_ref(x_out if hasF else xr, g2, b2)
        assert (st2[:, 0] - mean).abs().max().item() < 1e-4 and (st2[:, 1] - sd).abs().max().item() < 1e-4
        assert ((y2.float() - yr).abs() <= one_bf16(yr)).all()
        if n2:
            ref = y2.float() @ f(W["s2"]).t() + bs2[:n2 * d]
            assert ((out2.float() - ref).abs() <= one_bf16(ref) + 2e-3).all()
    # inference form (the decode's encoder pass): the tensors only a backward reads are not stored — hidden units, LayerNorm outputs,
    # statistics NULL — and the rest comes out bit for bit the same
    if hasF and has2:
        x_out_i, out2_i = torch.full_like(x_out, float("nan")), torch.zeros_like(out2)
        y2_keep = y2.clone()
        a.h, a.y1, a.st1, a.st2, a.x_out = None, None, None, None, x_out_i.data_ptr()
        if n2:
            a.y2, a.out2 = None, out2_i.data_ptr()
        L.check(L.lib().ortk_row_chain(C.byref(a), L.stream_ptr()), "ortk_row_chain (inference form)")
        torch.cuda.synchronize()
        assert torch.equal(x_out_i, x_out)
        if n2: assert torch.equal(out2_i, out2)
        else: assert torch.equal(y2, y2_keep)


def _ln_bwd_ref(gy, x, gain, eps=1e-6):
    n = x.size(-1)
    mean, sd = x.mean(-1, keepdim=True), x.std(-1, keepdim=True)
    r, xc = 1.0 / (sd + eps), x - mean
    g = gy * gain
    dx = r * (g - g.mean(-1, keepdim=True)) - r * r * (g * xc).sum(-1, keepdim=True) * xc / ((n - 1) * sd)
    return dx, (gy * xc * r).sum(0), gy.sum(0), torch.cat([mean, sd], 1)


@pytest.mark.parametrize("M,p,kind", [(1000, 0.1, "ZX"), (16640, 0.1, "ZX"), (9216, 0.0, "ZX"), (3000, 0.1, "X"), (2000, 0.1, "Y"), (777, 0.1, "Z")])
def test_row_bchain_vs_reference(L, M, p, kind):
    """ortk_row_bchain (the backward of the row-wise operators between two attention-backward calls in ONE rows-stationary launch:
    data gradients through the transposed weights, FFN gate, LayerNorm backward, masked bf16 copies) against the same chain in torch
    fp32 on the bf16-rounded operands with the kernels' own dropout masks.  kind: ZX = [dQKV . Wqkv -> LN' -> FFN' -> LN' -> . Wco]
    (the decoder's longest chain, 12 units), X = its tail on an existing masked gradient, Y = [dq . Wcq -> LN' -> . Wo], Z = the
    bottom of the stack ([dQKV . Wqkv -> LN'], no masked copy)."""
    d, NC = 512, 4
    ff = NC * d
    g = torch.Generator().manual_seed(M + 7)
    nin = {"ZX": 3, "X": 0, "Y": 1, "Z": 3}[kind]
    hasF, n2 = kind in ("ZX", "X"), (0 if kind == "Z" else 1)
    mk = lambda N, K: (torch.randn(N, K, generator=g) * (0.5 / math.sqrt(N))).bfloat16()
    Win = mk(nin * d, d) if nin else None                   # (nin*512 outputs, 512 inputs): dX = dY . W
    W2, W1, Wo = mk(d, ff), mk(ff, d), mk(d, d)
    # the transposed bf16 arena the chain streams (what cast_bf16_transposed leaves): block (N, K) stored as (K, N)
    parts, off, cur = [], {}, 0
    for name, W in (("in", Win), ("w2", W2), ("w1", W1), ("o", Wo)):
        if W is None: continue
        parts.append(W.t().contiguous().reshape(-1)); off[name] = cur; cur += W.numel()
    arena_t = dev(torch.cat(parts))
    units = [(off["in"] + i * d, nin * d) for i in range(nin)]
    if hasF:
        for c in range(NC):
            units += [(off["w2"] + c * d * d, d), (off["w1"] + c * d, ff)]
    if n2: units.append((off["o"], d))
    ut = dev(torch.tensor(units, dtype=torch.int64))
    ain = dev(rnd(M, max(nin, 1) * d, seed=1).bfloat16())
    dz0 = dev(rnd(M, d, seed=2).bfloat16())
    xa, xb = dev(rnd(M, d, seed=3)), dev(rnd(M, d, seed=4))
    ga, gb = dev(1 + rnd(d, seed=5, scale=0.1)), dev(1 + rnd(d, seed=6, scale=0.1))
    dresa, dresb_ext = dev(rnd(M, d, seed=7)), dev(rnd(M, d, seed=8))
    hrow = torch.relu(rnd(M, ff, seed=9)) * (torch.rand(M, ff, generator=g) < 0.9).float()        # post-ReLU, post-dropout hidden units
    hgate = dev(hrow.bfloat16())
    sta = torch.cat([xa.mean(1, keepdim=True), xa.std(1, keepdim=True)], 1).contiguous()
    stb = torch.cat([xb.mean(1, keepdim=True), xb.std(1, keepdim=True)], 1).contiguous()
    nanf = lambda *s: torch.full(s, float("nan"), device="cuda")
    dxa, dxb = nanf(M, d), nanf(M, d)
    daa, dba, dab, dbb = (torch.zeros(d, device="cuda") for _ in range(4))
    dza, dzb = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16), torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
    gh = torch.zeros(M, ff, device="cuda", dtype=torch.bfloat16)
    out2 = torch.zeros(M, d, device="cuda", dtype=torch.bfloat16)
    a = L.BChainArgs()
    a.w16t, a.units_dev, a.n_units = arena_t.data_ptr(), ut.data_ptr(), len(units)
    nb = L.lib().ortk_chain_packed_bytes(len(units))
    packed = torch.empty(nb, dtype=torch.uint8, device="cuda")
    a.packed, a.packed_bytes, a.M = packed.data_ptr(), nb, M
    sa, sb = 4321, 8765
    a.nin = nin
    if nin:
        a.ain, a.ld_ain = ain.data_ptr(), ain.stride(0)
        a.xa, a.sta, a.ga, a.dresa, a.dxa, a.daa, a.dba = xa.data_ptr(), sta.data_ptr(), ga.data_ptr(), dresa.data_ptr(), dxa.data_ptr(), daa.data_ptr(), dba.data_ptr()
        a.mask_a = 0 if kind == "Z" else 1
        a.dza, a.seed_a = dza.data_ptr(), sa
    else:
        a.dz0 = dz0.data_ptr()
    if hasF:
        a.NC, a.hgate, a.gh, a.gate_scale = NC, hgate.data_ptr(), gh.data_ptr(), 1.0 / (1.0 - p)
        dresb = dxa if nin else dresb_ext
        a.xb, a.stb, a.gb, a.dresb, a.dxb, a.dab, a.dbb = xb.data_ptr(), stb.data_ptr(), gb.data_ptr(), dresb.data_ptr(), dxb.data_ptr(), dab.data_ptr(), dbb.data_ptr()
        a.mask_b, a.dzb, a.seed_b = 1, dzb.data_ptr(), sb
    a.n2 = n2
    if n2: a.out2 = out2.data_ptr()
    a.drop_p, a.eps = p, 1e-6
    L.check(L.lib().ortk_row_bchain(C.byref(a), L.stream_ptr()), "ortk_row_bchain")
    torch.cuda.synchronize()
    # ---- reference
    f = lambda t: t.float().cuda()
    one_bf16 = lambda ref: 2.0 ** -7 * ref.abs().clamp(min=1.0)
    km = lambda seed: (_keep_mask(L, seed, M * d, p).view(M, d) if p > 0 else 1.0)
    if nin:
        gy = ain.float() @ f(Win)
        dx, da, db, _ = _ln_bwd_ref(gy, xa, ga)
        dx = dx + dresa
        assert (dxa - dx).abs().max().item() < 3e-3 * max(1.0, dx.abs().max().item())
        scale = max(1.0, da.abs().max().item())
        assert (daa - da).abs().max().item() < 2e-3 * scale and (dba - db).abs().max().item() < 2e-3 * max(1.0, db.abs().max().item())
        if kind == "Z":
            return
        zr = dxa * km(sa)                                    # (from the kernel's own dx: the products below take its rounded rows)
        assert ((dza.float() - zr).abs() <= one_bf16(zr)).all()
        dzin = dza.float()
    else:
        dzin = dz0.float()
    if hasF:
        ghr = (dzin @ f(W2)) * (hgate.float() > 0).float() * (1.0 / (1.0 - p))
        assert ((gh.float() - ghr).abs() <= one_bf16(ghr) + 2e-3).all()
        gy2 = gh.float() @ f(W1)
        dx, da, db, _ = _ln_bwd_ref(gy2, xb, gb)
        dx = dx + (dxa if nin else dresb_ext)
        assert (dxb - dx).abs().max().item() < 3e-3 * max(1.0, dx.abs().max().item())
        assert (dab - da).abs().max().item() < 2e-3 * max(1.0, da.abs().max().item()) and (dbb - db).abs().max().item() < 2e-3 * max(1.0, db.abs().max().item())
        zr = dxb * km(sb)
        assert ((dzb.float() - zr).abs() <= one_bf16(zr)).all()
        dzin = dzb.float()
    ref = dzin @ f(Wo)
    assert ((out2.float() - ref).abs() <= one_bf16(ref) + 2e-3).all()
