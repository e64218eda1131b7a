"""Row-panel GEMM with the LayerNorm in its epilogue (ortk_gemm ln_mode 1 / 2) against the separate launches, in isolation."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_image_captioning_amd as P

L = P._lib
L.require_gpu()


def args(A, W, Cout, M, K, **kw):
    a = L.GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), W.data_ptr(), Cout.data_ptr()
    a.lda, a.ldb, a.ldc = K, K, 512
    a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype = M, 512, K, 1, 1, 1
    for k, v in kw.items():
        setattr(a, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return a


def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M in (16640, 9216, 21760):
    for K in (512, 1536, 2048):
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(512, K, device="cuda") * K ** -0.5).bfloat16()
        bias = torch.randn(512, device="cuda"); res = torch.randn(M, 512, device="cuda")
        ga = torch.ones(512, device="cuda"); gb = torch.zeros(512, device="cuda")
        c = torch.empty(M, 512, device="cuda"); y = torch.empty(M, 512, device="cuda", dtype=torch.bfloat16); st = torch.empty(M, 2, device="cuda")
        x = torch.randn(M, 512, device="cuda"); dres = torch.randn(M, 512, device="cuda"); dx = torch.empty(M, 512, device="cuda")
        da = torch.zeros(512, device="cuda"); db = torch.zeros(512, device="cuda")
        s = L.stream_ptr()
        lib = L.lib()
        a_plain = args(A, W, c, M, K, bias=bias, resid=res, ldr=512, drop_p=0.1, drop_seed=3)
        a_f1 = args(A, W, c, M, K, bias=bias, resid=res, ldr=512, drop_p=0.1, drop_seed=3, ln_mode=1, ln_y_dtype=1, ln_a=ga, ln_b=gb, ln_y=y, ln_stats=st, ln_eps=1e-6)
        a_d = args(A, W, c, M, K)
        a_f2 = args(A, W, dx, M, K, ln_mode=2, ln_y_dtype=1, ln_a=ga, ln_y=y, ln_stats=st, ln_eps=1e-6, ln_x=x, ln_dres=dres, ln_da=da, ln_db=db, drop_p=0.1, drop_seed=5)
        lib.ortk_layernorm_fwd(L.ptr(x), L.ptr(ga), L.ptr(gb), L.ptr(y), 1, L.ptr(st), M, 512, 1e-6, s)

        def sep1():
            lib.ortk_gemm(C.byref(a_plain), s)
            lib.ortk_layernorm_fwd(L.ptr(c), L.ptr(ga), L.ptr(gb), L.ptr(y), 1, L.ptr(st), M, 512, 1e-6, s)

        def sep2():
            lib.ortk_gemm(C.byref(a_d), s)
            lib.ortk_layernorm_bwd_drop(L.ptr(c), L.ptr(x), L.ptr(ga), L.ptr(st), L.ptr(dres), L.ptr(dx), L.ptr(da), L.ptr(db), M, 512, 1e-6, L.ptr(y), 1, 0.1, 5, s)

        t_g = timeit(lambda: lib.ortk_gemm(C.byref(a_plain), s))
        t_s1 = timeit(sep1); t_f1 = timeit(lambda: lib.ortk_gemm(C.byref(a_f1), s))
        lib.ortk_layernorm_fwd(L.ptr(x), L.ptr(ga), L.ptr(gb), L.ptr(y), 1, L.ptr(st), M, 512, 1e-6, s)
        t_s2 = timeit(sep2); t_f2 = timeit(lambda: lib.ortk_gemm(C.byref(a_f2), s))
        print(f"M {M:6d} K {K:5d}: gemm alone {t_g:6.1f} us | fwd  gemm+ln {t_s1:6.1f} fused {t_f1:6.1f} | bwd  gemm+ln_bwd {t_s2:6.1f} fused {t_f2:6.1f}", flush=True)
