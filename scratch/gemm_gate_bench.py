"""The gated data-gradient GEMM of the feed-forward sublayer alone (rows x 2 048 x 512, bf16 result, bf16 gate = the hidden units):
lean epilogue against the general one (ortk_tuning.gemm_epilogue bit 0), us per launch."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L = P._lib; lib = L.lib()
def timeit(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for M in (16640, 9216, 21760):
    N, K = 2048, 512
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    gate = torch.randn(M, N, device="cuda").bfloat16(); Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"M {M} N {N} K {K}:"
    for general in (1, 0, 1, 0):
        L.set_tuning(gemm_epilogue=general)
        a = L.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, 1
        a.gate, a.ldg, a.gate_dtype, a.gate_scale = gate.data_ptr(), N, 1, 1.0 / 0.9
        t = timeit(lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr()))
        line += f"  {'general' if general else 'lean'} {t:6.1f}"
    print(line + "  us", flush=True)
L.set_tuning(gemm_epilogue=0)
