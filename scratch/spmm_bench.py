"""Microbenchmark: ortk_spmm_csr vs the dense bf16 / fp32 GEMM on the decode- and training-sized projections."""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import csr_from_dense
L = P._lib; lib = L.lib()

def t_ms(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for sp in (0.95, 0.99):
    for (M, N, K) in [(5120, 512, 512), (5120, 1536, 512), (5120, 2048, 512), (5120, 512, 2048), (36864, 512, 512), (36864, 2048, 512), (36864, 512, 2048), (5120, 10240, 512)]:
        W = torch.randn(N, K, device="cuda") * (torch.rand(N, K, device="cuda") >= sp).float()
        rp, col, val = csr_from_dense(W)
        csr = L.Csr(L.ptr(rp), L.ptr(col), L.ptr(val), N, K, 0)
        for xdt, name in ((1, "bf16"), (0, "fp32")):
            X = torch.randn(M, K, device="cuda").to(torch.bfloat16 if xdt else torch.float32)
            Y = torch.empty(M, N, device="cuda")
            bias = torch.randn(N, device="cuda")
            f_sp = lambda: lib.ortk_spmm_csr(C.byref(csr), L.ptr(X), xdt, K, L.ptr(bias), L.ptr(Y), 0, N, M, 0, None, 0, L.stream_ptr())
            Wd = W.to(torch.bfloat16 if xdt else torch.float32)
            a = L.GemmArgs(); a.A, a.B, a.C = L.ptr(X), L.ptr(Wd), L.ptr(Y); a.lda, a.ldb, a.ldc = K, K, N
            a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype = M, N, K, xdt, xdt, xdt; a.bias = L.ptr(bias)
            f_d = lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr())
            ts, td = t_ms(f_sp), t_ms(f_d)
            print(f"sp={sp} M={M} N={N} K={K} {name}: spmm {ts*1e3:8.1f} us   dense {td*1e3:8.1f} us   ratio {td/ts:5.2f}", flush=True)
