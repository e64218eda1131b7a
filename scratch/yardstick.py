"""Yardstick: this library's bf16 MFMA GEMM kernels vs hipBLASLt (through torch.matmul; never linked into the product) on
every GEMM shape of one XE step (B = 256) in the three operand layouts.  Interleaved rounds in one process."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L = P._lib

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

def ortk(M, N, K, ta, tb, A, B, Cc, acc=0, splitk=1):
    a = L.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), N
    a.M, a.N, a.K, a.transA, a.transB, a.precision = M, N, K, ta, tb, 1; a.accumulate = acc; a.splitk = splitk; a.a_dtype = 1; a.b_dtype = 1
    a.c_dtype = 0 if acc else 1
    return lambda: L.lib().ortk_gemm(C.byref(a), L.stream_ptr())

def splitk(M, Nout, Kin):
    tiles = ((Nout + 127) // 128) * ((Kin + 127) // 128)
    sk = (3 if M >= 16384 else 1) if tiles >= 256 else (384 + tiles // 2) // tiles
    return max(1, min(sk, max(1, M // 512)))

Me, Md = 9216, 21760
shapes = [("att", Me, 512, 2048, 1), ("e.qkv", Me, 1536, 512, 6), ("e.wo", Me, 512, 512, 6), ("e.w1", Me, 2048, 512, 6), ("e.w2", Me, 512, 2048, 6),
          ("ckv", Me, 6144, 512, 1), ("d.qkv", Md, 1536, 512, 6), ("d.wo/cq/co", Md, 512, 512, 18), ("d.w1", Md, 2048, 512, 6), ("d.w2", Md, 512, 2048, 6),
          ("gen", Md, 10112, 512, 1)]
tot = {}
for kind in ("fwd", "dgrad", "wgrad"):
    for name, M, N, K, cnt in shapes:
        if kind == "dgrad" and name == "att":
            continue
        bf = torch.bfloat16
        if kind == "fwd":          # Y (M,N) = X (M,K) W^T, W (N,K)
            X, W = torch.randn(M, K, device="cuda").to(bf), torch.randn(N, K, device="cuda").to(bf)
            Y = torch.empty(M, N, device="cuda", dtype=bf)
            f1 = ortk(M, N, K, 0, 0, X, W, Y); Wt = W.t()
            f2 = lambda: torch.matmul(X, Wt, out=Y)
            flops = 2 * M * N * K
        elif kind == "dgrad":      # dX (M,K) = dY (M,N) W; the product uses the transposed bf16 copy W^T (K,N): forward layout
            dY, WT = torch.randn(M, N, device="cuda").to(bf), torch.randn(K, N, device="cuda").to(bf)
            dX = torch.empty(M, K, device="cuda", dtype=bf)
            f1 = ortk(M, K, N, 0, 0, dY, WT, dX); WTt = WT.t()
            f2 = lambda: torch.matmul(dY, WTt, out=dX)
            flops = 2 * M * N * K
        else:                      # dW (N,K) += dY^T (N,M) X (M,K): both operands stored row-major over the M rows
            dY, X = torch.randn(M, N, device="cuda").to(bf), torch.randn(M, K, device="cuda").to(bf)
            dW = torch.zeros(N, K, device="cuda"); dW16 = torch.empty(N, K, device="cuda", dtype=bf)
            f1 = ortk(N, K, M, 1, 1, dY, X, dW, acc=1, splitk=splitk(M, N, K)); dYt = dY.t()
            f2 = lambda: torch.matmul(dYt, X, out=dW16)
            flops = 2 * M * N * K
        t1, t2 = t_us([f1, f2])
        tot[kind] = tot.get(kind, (0, 0))
        tot[kind] = (tot[kind][0] + t1 * cnt, tot[kind][1] + t2 * cnt)
        print(f"{kind:5s} {name:10s} M{M:6d} N{N:6d} K{K:5d} x{cnt:2d}: ortk {t1:7.1f} us {flops/t1/1e6:6.0f} TF/s | hipblaslt {t2:7.1f} us {flops/t2/1e6:6.0f} TF/s | ortk/hipblaslt time {t1/t2:5.2f}", flush=True)
for k, (a, b) in tot.items():
    print(f"per step {k}: ortk {a/1e3:.3f} ms   hipblaslt {b/1e3:.3f} ms   ratio {a/b:.2f}")
