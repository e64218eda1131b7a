"""Microbenchmark: ortk_spmm_ell (sorted-ELL sparse product) vs the dense bf16 MFMA GEMM of this library on zero-filled
weights vs hipBLASLt through torch.matmul (yardstick only, never linked into the product), on the decode- and
training-sized projections.  Interleaved rounds in one process (cdna guide rule 24)."""
import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import EllPlan, capacity_for
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
          (9216, 512, 2048), (9216, 1536, 512), (21760, 512, 512), (21760, 1536, 512), (21760, 2048, 512), (21760, 512, 2048), (21760, 10112, 512)]
for sp in (0.95, 0.975):
    for (M, N, K) in shapes:
        W = torch.randn(N, K, device="cuda") * (torch.rand(N, K, device="cuda") >= sp).float()
        plan = EllPlan([dict(offset=0, N=N, K=K, ld=K, capacity=capacity_for(N, K, 0.1))], 4, "cuda")
        W16 = W.bfloat16()
        plan.build(W16); plan.check_overflow()
        X = torch.randn(M, K, device="cuda").bfloat16()
        Y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        sa = L.SpmmArgs(); sa.X, sa.Y, sa.ldx, sa.ldy, sa.M, sa.x_dtype, sa.y_dtype, sa.bias = X.data_ptr(), Y.data_ptr(), K, N, M, 1, 1, bias.data_ptr()
        f_sp = lambda: lib.ortk_spmm_ell(plan.ref(), 0, C.byref(sa), L.stream_ptr())
        a = L.GemmArgs(); a.A, a.B, a.C = L.ptr(X), L.ptr(W16), L.ptr(Y); a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, 1; a.bias = L.ptr(bias)
        f_d = lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr())
        Wt = W16.t().contiguous()
        f_blas = lambda: torch.matmul(X, Wt, out=Y)
        f_build = lambda: plan.build(W16)
        ts, td, tb, tbu = t_us([f_sp, f_d, f_blas, f_build])
        nnz = int((W != 0).sum())
        algo = 2 * M * K + 2 * M * N + 4 * nnz
        print(f"sp={sp} M={M:6d} N={N:6d} K={K:5d}: spmm {ts:7.1f} us  dense(ortk) {td:7.1f} us  hipblaslt {tb:7.1f} us  build {tbu:6.1f} us | "
              f"dense/spmm {td/ts:5.2f}  | spmm {M*nnz/ts/1e6:6.1f} Gprod/ms-> {2*M*nnz/ts/1e6:7.1f} TFLOP/s-eq  algoHBM {algo/ts/1e3:7.1f} GB/s", flush=True)
