"""What does the decode-time generator pay for storing its logits?  (generator + beam step without the 207 MB fp32
store per position.)  Generator GEMM of one decode position — rows x 10 112 x 512, bf16 operands, bias, soft-max partials epilogue
(tile_stats) — with the logits stored as fp32 (what the decode does) and as bf16 (half the bytes): the difference bounds what a
store-free epilogue could save."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sparse_image_captioning_amd as pkg

L = pkg._lib
L.require_gpu()
lib = L.lib()
N, K, V = 10112, 512, 10001
for M in (5120, 1536, 16640):      # decode (1 024 images x 5 beams), SCST rollout (256 x 6), the XE step's valid positions
    A = (torch.randn(M, K, device="cuda") * 0.5).bfloat16()
    B = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    bias = torch.randn(N, device="cuda")
    stats = torch.empty(M, N // 64, 2, device="cuda")
    for cdt, name in ((0, "fp32 logits"), (1, "bf16 logits")):
        Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16 if cdt else torch.float32)
        a = L.GemmArgs()
        a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cc.data_ptr()
        a.lda, a.ldb, a.ldc = K, K, N
        a.M, a.N, a.K, a.precision = M, N, K, 1
        a.a_dtype, a.b_dtype, a.c_dtype = 1, 1, cdt
        a.bias = bias.data_ptr()
        for st in (True, False):
            a.tile_stats, a.stat_ncols = (stats.data_ptr(), V) if st else (None, 0)
            for _ in range(5):
                L.check(lib.ortk_gemm(C.byref(a), L.stream_ptr()), "gemm")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(100):
                L.check(lib.ortk_gemm(C.byref(a), L.stream_ptr()), "gemm")
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 10.0
            print(f"M {M:5d}  {name}  stats {int(st)}  {us:7.1f} us   {2.0 * M * N * K / us / 1e6:6.1f} TF/s   C bytes {Cc.numel() * Cc.element_size() / 1e6:6.1f} MB")
