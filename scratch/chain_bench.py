"""The rows-stationary chain kernel (ortk_row_chain) against the separate kernels it replaces, on the decoder's
[Wco -> +x -> LN -> W1 .. W2 -> +x -> LN -> Wqkv] chain (13 units) and the [Wo -> +x -> LN -> Wcq] chain (2 units), at the row counts of the
path.  python scratch/chain_bench.py"""
import ctypes as C, math, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
L = P._lib; lib = L.lib()
d, NC = 512, 4; ff = NC * d


def t_us(fns, n=10, rounds=3):
    best = [1e9] * len(fns)
    for f in fns:
        for _ in range(2): f()
    for _ in range(rounds):
        for i, f in enumerate(fns):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize()
            best[i] = min(best[i], a.elapsed_time(b) / n * 1e3)
    return best


def gemm(A, B, Cc, M, N, K, **kw):
    a = L.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), Cc.stride(0)
    a.M, a.N, a.K, a.precision = M, N, K, 1
    a.a_dtype, a.b_dtype, a.c_dtype = 1, 1, (1 if Cc.dtype == torch.bfloat16 else 0)
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            setattr(a, k, v.data_ptr())
            if k == "resid": a.ldr = v.stride(0)
        else: setattr(a, k, v)
    return a


PF = len(sys.argv) < 2 or sys.argv[1] not in ("nopf", "wide", "narrow")
if len(sys.argv) > 1 and sys.argv[1] in ("wide", "narrow"):            # no prefetchers; the 76-row form where it saves a round | never
    L.set_tuning(chain_wide=1 if sys.argv[1] == "wide" else 0)
for M in (5120, 9216, 16640, 21760, 36864):
    g = torch.Generator().manual_seed(1)
    Wr = (torch.randn(d, d, generator=g) * 0.02).bfloat16().cuda(); W1 = (torch.randn(ff, d, generator=g) * 0.02).bfloat16().cuda()
    W2 = (torch.randn(d, ff, generator=g) * 0.02).bfloat16().cuda(); Ws = (torch.randn(3 * d, d, generator=g) * 0.02).bfloat16().cuda()
    arena = torch.cat([Wr.reshape(-1), W1.reshape(-1), W2.reshape(-1), Ws.reshape(-1)])
    oR, o1, o2, oS = 0, d * d, d * d + ff * d, d * d + 2 * ff * d
    Wr, W1, W2, Ws = arena[oR:o1].view(d, d), arena[o1:o2].view(ff, d), arena[o2:oS].view(d, ff), arena[oS:].view(3 * d, d)
    x = torch.randn(M, d, device="cuda"); a_in = torch.randn(M, d, device="cuda").bfloat16()
    bias = torch.randn(4 * d, device="cuda") * 0.1; gam = torch.ones(d, device="cuda")
    x_mid, x_out = torch.empty(M, d, device="cuda"), torch.empty(M, d, device="cuda")
    y1, y2 = torch.empty(M, d, device="cuda", dtype=torch.bfloat16), torch.empty(M, d, device="cuda", dtype=torch.bfloat16)
    st1, st2 = torch.empty(M, 2, device="cuda"), torch.empty(M, 2, device="cuda")
    h = torch.empty(M, ff, device="cuda", dtype=torch.bfloat16); qkv = torch.empty(M, 3 * d, device="cuda", dtype=torch.bfloat16)
    qc = torch.empty(M, d, device="cuda", dtype=torch.bfloat16)

    def chain(full):
        units = [(oR, d)] + ([] if full else [(oS, d)])
        if full:
            for c in range(NC): units += [(o1 + c * d * d, d), (o2 + c * d, ff)]
            units += [(oS + i * d * d, d) for i in range(3)]
        ut = torch.tensor(units, dtype=torch.int64).cuda()
        a = L.ChainArgs(); a.w16, a.units_dev, a.n_units = arena.data_ptr(), ut.data_ptr(), len(units)
        nb = lib.ortk_chain_packed_bytes(len(units)); packed = torch.empty(nb, dtype=torch.uint8, device="cuda")
        a.packed, a.packed_bytes, a.M, a.x_in = packed.data_ptr(), nb, M, x.data_ptr()
        a.a_in, a.bias_r, a.x_mid, a.seed_r = a_in.data_ptr(), bias.data_ptr(), x_mid.data_ptr(), 1
        a.g1, a.b1, a.y1, a.st1 = gam.data_ptr(), bias.data_ptr(), y1.data_ptr(), st1.data_ptr()
        if full:
            a.NC, a.bias_h, a.bias_o, a.h, a.x_out, a.seed_h, a.seed_o = NC, bias.data_ptr(), bias.data_ptr(), h.data_ptr(), x_out.data_ptr(), 2, 3
            a.g2, a.b2, a.y2, a.st2 = gam.data_ptr(), bias.data_ptr(), y2.data_ptr(), st2.data_ptr()
            a.n2, a.bias_s2, a.out2, a.ld2 = 3, bias.data_ptr(), qkv.data_ptr(), 3 * d
        else:
            a.n1, a.bias_s1, a.out1, a.ld1 = 1, bias.data_ptr(), qc.data_ptr(), d
        a.drop_p, a.eps = 0.1, 1e-6
        prog = torch.zeros(16, dtype=torch.int32, device="cuda")
        if PF: a.progress = prog.data_ptr()
        keep = (ut, packed, prog)
        lib.ortk_row_chain(C.byref(a), L.stream_ptr())            # (packs)
        from sparse_image_captioning_amd import _lib
        return a, keep

    aF, kF = chain(True); aB, kB = chain(False)
    # the chain alone, on a prepacked stream (what the executor does: one pack per forward): time ortk_row_chain minus the pack
    fF = lambda: lib.ortk_row_chain(C.byref(aF), L.stream_ptr())
    fB = lambda: lib.ortk_row_chain(C.byref(aB), L.stream_ptr())
    # separate kernels
    gR = gemm(a_in, Wr, x_mid, M, d, d, bias=bias, resid=x, drop_p=0.1, drop_seed=1)
    g1 = gemm(y1, W1, h, M, ff, d, bias=bias, relu=1, drop_p=0.1, drop_seed=2)
    g2 = gemm(h, W2, x_out, M, d, ff, bias=bias, resid=x_mid, drop_p=0.1, drop_seed=3)
    gS = gemm(y2, Ws, qkv, M, 3 * d, d, bias=bias)
    gQ = gemm(y1, Ws, qc, M, d, d, bias=bias)
    ln = lambda xx, yy, ss: lib.ortk_layernorm_fwd(L.ptr(xx), L.ptr(gam), L.ptr(bias), L.ptr(yy), 1, L.ptr(ss), M, d, 1e-6, L.stream_ptr())

    def sepF():
        lib.ortk_gemm(C.byref(gR), L.stream_ptr()); ln(x_mid, y1, st1); lib.ortk_gemm(C.byref(g1), L.stream_ptr())
        lib.ortk_gemm(C.byref(g2), L.stream_ptr()); ln(x_out, y2, st2); lib.ortk_gemm(C.byref(gS), L.stream_ptr())

    def sepB():
        lib.ortk_gemm(C.byref(gR), L.stream_ptr()); ln(x_mid, y1, st1); lib.ortk_gemm(C.byref(gQ), L.stream_ptr())
    tF, tsF, tB, tsB = t_us([fF, sepF, fB, sepB])
    print(f"M={M:6d}: [Wo LN W1 W2 LN Wqkv] chain+pack {tF:7.1f} us  separate {tsF:7.1f} us | [Wo LN Wcq] chain+pack {tB:6.1f} us  separate {tsB:6.1f} us", flush=True)
