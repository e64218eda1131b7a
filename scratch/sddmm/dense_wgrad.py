"""The library's dense weight-gradient GEMM (dW = dY^T X, bf16 operands, split-K + fp32 atomics) on the shape of
scratch/sddmm/sddmm_probe.hip, + the ortk_mask_bwd pass a binary-masked layer adds: what an SDDMM would have to beat."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sparse_image_captioning_amd as pkg
L = pkg._lib
lib = L.lib()
M, N, K = 21760, 512, 512
dY = (torch.randn(M, N, device="cuda") * 0.02).bfloat16()
X = (torch.randn(M, K, device="cuda") * 0.02).bfloat16()
dW = torch.zeros(N, K, device="cuda")
a = L.GemmArgs()
a.A, a.B, a.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
a.lda, a.ldb, a.ldc = N, K, K
a.M, a.N, a.K, a.transA, a.transB = N, K, M, 1, 1
a.precision, a.a_dtype, a.b_dtype, a.c_dtype, a.accumulate = 1, 1, 1, 0, 1
a.splitk = 24          # the executor's choice for this shape (ortk_model.hip: wgrad_gemm: ~384 workgroups)
for _ in range(3):
    L.check(lib.ortk_gemm(C.byref(a), L.stream_ptr()), "ortk_gemm")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    lib.ortk_gemm(C.byref(a), L.stream_ptr())
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print(f"dense wgrad {N} x {K} over M = {M}: {us:7.1f} us = {2.0 * M * N * K / (us * 1e-6) / 1e12:6.1f} TFLOP/s (+ the masking pass over the arena: one launch per step for all layers)")
ref = dY.float().t() @ X.float()
print("check", ((dW / 53 - ref).abs().max() / ref.abs().max()).item())
