"""Weight-gradient GEMM (dW += dY^T X, fused bias gradient) over the path's shapes with the split-K rule of the executor:
run once per ORTK_GEMM_IMPL setting (0 = register-staged kernel, 2 = LDS-DMA 128 x 128 tiles) and compare."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sparse_image_captioning_amd as P
import os as _os
if "ORTK_GEMM_IMPL" in _os.environ:      # (the library itself reads no environment: forward the old switch through ortk_set_tuning)
    P._lib.set_tuning(gemm_impl=int(_os.environ["ORTK_GEMM_IMPL"]))

L = P._lib; L.require_gpu()
tot = 0.0
for rows in (16640, 9216):
    for (No, Ki, cnt) in ((512, 512, 3 if rows > 10000 else 1), (1536, 512, 1), (2048, 512, 1), (512, 2048, 1)):
        dY = torch.randn(rows, No, device="cuda").bfloat16(); X = torch.randn(rows, Ki, device="cuda").bfloat16()
        dW = torch.zeros(No, Ki, device="cuda"); db = torch.zeros(No, device="cuda")
        a = L.GemmArgs(); a.A, a.B, a.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr(); a.lda, a.ldb, a.ldc = No, Ki, Ki
        a.M, a.N, a.K, a.transA, a.transB, a.precision, a.accumulate, a.a_dtype, a.b_dtype = No, Ki, rows, 1, 1, 1, 1, 1, 1
        tiles = (No // 128) * (Ki // 128)
        sk = (3 if rows >= 16384 else 1) if tiles >= 256 else (384 + tiles // 2) // tiles
        a.splitk = max(1, min(sk, rows // 512)); a.colsum = db.data_ptr()
        for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a), L.stream_ptr()), "g")
        torch.cuda.synchronize()
        ref = dY.float().t() @ X.float(); refb = dY.float().sum(0)
        dW.zero_(); db.zero_(); L.lib().ortk_gemm(C.byref(a), L.stream_ptr()); torch.cuda.synchronize()
        err = (dW - ref).abs().max().item() / ref.abs().max().item(); errb = (db - refb).abs().max().item() / refb.abs().max().item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): L.lib().ortk_gemm(C.byref(a), L.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        tot += us * cnt * 6
        print(f"rows {rows:6d} dW {No:5d} x {Ki:5d} splitk {a.splitk:3d}: {us:7.1f} us {2.0 * No * Ki * rows / us / 1e6:7.1f} TF/s  err {err:.1e} bias err {errb:.1e}", flush=True)
print("per step (6 layers each): %.2f ms" % (tot / 1e3))
