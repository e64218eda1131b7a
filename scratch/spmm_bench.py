"""Microbenchmark: ortk_spmm (GU16 group-union MFMA product, and the ELL16 VALU product for comparison) vs the dense bf16 MFMA
GEMM of this library on zero-filled weights vs hipBLASLt through torch.matmul (yardstick only, never linked into the
product), on the decode- and training-sized projections.  Interleaved rounds in one process (cdna guide rule 24)."""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
L = P._lib; lib = L.lib()

def t_us(fns, n=20, rounds=3):
    best = [1e9] * len(fns)
    for f in fns:
        for _ in range(3): f()
    for _ in range(rounds):
        for i, f in enumerate(fns):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize()
            best[i] = min(best[i], a.elapsed_time(b) / n * 1e3)
    return best

shapes = [(5120, 512, 512), (5120, 1536, 512), (5120, 2048, 512), (5120, 512, 2048), (5120, 10112, 512),
          (9216, 512, 2048), (9216, 1536, 512), (21760, 512, 512), (21760, 1536, 512), (21760, 2048, 512), (21760, 512, 2048), (21760, 10112, 512),
          (21760, 512, 10112)]
for sp in (0.95, 0.975):
    for (M, N, K) in shapes:
        W = torch.randn(N, K, device="cuda") * (torch.rand(N, K, device="cuda") >= sp).float()
        W16 = W.bfloat16()
        gu = SparsePlan([dict(offset=0, N=N, K=K, ld=K)], L.SP_GU16, "cuda")
        gu.build(W16)
        ell = None
        if K <= 2048:
            ell = SparsePlan([dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 0.1))], L.SP_ELL16, "cuda")
            ell.build(W16); ell.check_overflow()
        X = torch.randn(M, K, device="cuda").bfloat16()
        Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        sa = L.SpmmArgs(); sa.X, sa.Y, sa.ldx, sa.ldy, sa.M, sa.x_dtype, sa.y_dtype, sa.bias = X.data_ptr(), Y.data_ptr(), K, N, M, 1, 1, bias.data_ptr()
        f_gu = lambda: lib.ortk_spmm(gu.ref(), 0, C.byref(sa), L.stream_ptr())
        f_ell = (lambda: lib.ortk_spmm(ell.ref(), 0, C.byref(sa), L.stream_ptr())) if ell else (lambda: None)
        a = L.GemmArgs(); a.A, a.B, a.C = L.ptr(X), L.ptr(W16), L.ptr(Y); a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, 1; a.bias = L.ptr(bias)
        f_d = lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr())
        Wt = W16.t().contiguous()
        f_blas = lambda: torch.matmul(X, Wt, out=Y)
        f_build = lambda: gu.build(W16)
        tg, te, td, tb, tbu = t_us([f_gu, f_ell, f_d, f_blas, f_build])
        nnz = int((W != 0).sum())
        steps = int(gu.chunk_len.sum())
        algo = 2 * M * K + 2 * M * N + 4 * nnz
        print(f"sp={sp} M={M:6d} N={N:6d} K={K:5d}: gu16 {tg:7.1f} us  ell16 {te:7.1f}  dense(ortk) {td:7.1f}  hipblaslt {tb:7.1f}  build {tbu:6.1f} | "
              f"dense/gu16 {td/tg:5.2f} | union K'/K {steps*32/((N+15)//16)/K:4.2f}  mfma {2*M*16*steps*32/tg/1e6:7.1f} TF/s  algoHBM {algo/tg/1e3:7.1f} GB/s", flush=True)
