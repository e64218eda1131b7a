"""Fixed cost of a forward-layout GEMM launch at the XE step's 16 640 decoder rows: us against K, fp32 and bf16 results, with the lean
epilogue (ortk_tuning.gemm_epilogue = 0) and the general one (1).  -> profiles/r06_gemm_epilogue.txt"""
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
M = 16640
for N in (512, 2048):
    for cdt, general in ((0, 1), (0, 0), (1, 1), (1, 0)):
        L.set_tuning(gemm_epilogue=general)
        line = f"M {M} N {N} out {'bf16' if cdt else 'fp32'} {'general' if general else 'lean   '} epilogue:"
        for K in (64, 128, 256, 512, 1024, 2048):
            A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
            Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if cdt else torch.float32)
            a = L.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb, a.ldc = K, K, N
            a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, cdt
            t = timeit(lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr()))
            line += f"  K={K}: {t:5.1f}"
        print(line + "  us", flush=True)
