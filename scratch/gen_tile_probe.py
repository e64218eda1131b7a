"""Decode-time generator GEMM (5 120 rows x 10 240 padded vocabulary x 512, bf16 operands, fp32 logits + bias): 128 x 128 LDS-DMA tiles
(what the soft-max-partials epilogue lives in) against the 256 x 256 tiles, without the partials — is porting the epilogue worth it?
    python scratch/gen_tile_probe.py"""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_image_captioning_amd import _lib as L   # noqa: E402
from f32x3_bench import timeit   # noqa: E402

M, N, K = 5120, 10240, 512
g = torch.Generator().manual_seed(1)
A = torch.randn(M, K, generator=g).cuda().bfloat16(); B = (torch.randn(N, K, generator=g) * 0.05).cuda().bfloat16()
bias = torch.randn(N, generator=g).cuda()
Cout = torch.empty(M, N, device="cuda")
stats = torch.empty(M, N // 64, 2, device="cuda")


def gemm(with_stats):
    a = L.GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    a.lda, a.ldb, a.ldc = K, K, N
    a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype = M, N, K, 1, 1, 1
    a.bias = bias.data_ptr()
    if with_stats:
        a.tile_stats = stats.data_ptr(); a.stat_ncols = 10112
    L.check(L.lib().ortk_gemm(C.byref(a), L.stream_ptr()), "ortk_gemm")


for impl, name in ((0, "automatic (256 x 256 tiles)"), (2, "128 x 128 tiles")):
    L.set_tuning(gemm_impl=impl)
    print(f"{name:30s}: plain {timeit(lambda: gemm(False), 50):6.1f} us   with partials {timeit(lambda: gemm(True), 50):6.1f} us", flush=True)
L.set_tuning(gemm_impl=0)
