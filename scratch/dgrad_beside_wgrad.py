"""What slows a data-gradient GEMM of the XE step down beside a grouped weight-gradient launch on another stream?
dgrad: (16640 x 512) x (512 x 512)^T on the 256 x 256 LDS-DMA tiles (130 workgroups), repeated back to back on stream 1 while
stream 2 runs decoder-layer weight-gradient groups of a given shape (workgroups = 56 x splitk)."""
import sys, ctypes as C, time
sys.path[:0] = ["/root/repo", "/root/repo/scratch"]
import torch
from wgrad_group_bench import group, L, lib
d, ff = 512, 2048
M = 16640
def mk_dgrad(N=512, K=512, out_f32=True):
    A = torch.randn(M, K, device="cuda").bfloat16(); B = torch.randn(N, K, device="cuda").bfloat16()
    Cc = torch.zeros(M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    a = L.GemmArgs(); a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr(); a.lda, a.ldb, a.ldc = K, K, N
    a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, 0 if out_f32 else 1
    return a, (A, B, Cc)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
shapes = [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]
def run(dg, wg, n_d=60, n_w=0):
    """time of n_d dgrads on s1 while s2 (optionally) loops wgrad groups"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(n_w): lib.ortk_wgrad_group(C.byref(wg), s2.cuda_stream)
    e0.record(s1)
    for _ in range(n_d): lib.ortk_gemm(C.byref(dg), s1.cuda_stream)
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n_d * 1e3
for (N, K, f32) in ((512, 512, True), (2048, 512, False), (512, 2048, True)):
    dg, keep = mk_dgrad(N, K, f32)
    run(dg, None, 5)
    base = run(dg, None)
    line = f"dgrad {M}x{N}x{K}: alone {base:6.1f} us"
    for sk, fl, name in ((1, 0, "56 wgs"), (2, 0, "112 wgs"), (4, 0, "224 wgs")):
        wg, keep2 = group(M, shapes, sk); wg.flags = fl
        lib.ortk_wgrad_group(C.byref(wg), s2.cuda_stream); torch.cuda.synchronize()
        # enough wgrad launches to cover the dgrads
        t = run(dg, wg, 60, 40 if sk == 1 else 60)
        line += f" | beside {name}: {t:6.1f}"
    print(line, flush=True)
