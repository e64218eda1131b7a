"""fp32 data-gradient / weight-gradient GEMMs of one XE step (16 640 decoder rows): fp32 MFMA kernels (f32_split = 0) against the split
kernels (gemm_f32x3t_kernel), us per launch and fp32-equivalent TF/s.   python scratch/f32x3t_bench.py [variants...]"""
import ctypes as C
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparse_image_captioning_amd import _lib as L   # noqa: E402
from f32x3_bench import timeit   # noqa: E402


def gemm(A, B, Cout, M, N, K, ta, tb, acc=0, splitk=0):
    a = L.GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), Cout.stride(0)
    a.M, a.N, a.K, a.transA, a.transB, a.precision, a.accumulate, a.splitk = M, N, K, ta, tb, 0, acc, splitk
    L.check(L.lib().ortk_gemm(C.byref(a), L.stream_ptr()), "ortk_gemm")


variants = [int(v) for v in sys.argv[1:]] or [0, 1]
R = 16640
g = torch.Generator().manual_seed(1)
for name, nout, kin in (("qkv", 1536, 512), ("wo/cq/co", 512, 512), ("w1", 2048, 512), ("w2", 512, 2048), ("gen", 10112, 512)):
    dY = torch.randn(R, nout, generator=g).cuda(); W = (torch.randn(nout, kin, generator=g) * 0.05).cuda(); X = torch.randn(R, kin, generator=g).cuda()
    dX = torch.empty(R, kin, device="cuda"); dW = torch.zeros(nout, kin, device="cuda")
    tiles = -(-nout // 128) * -(-kin // 128)
    sk = (3 if R >= 16384 else 1) if tiles >= 256 else max(1, min((384 + tiles // 2) // tiles, R // 512))      # ortk_model.hip: wgrad_gemm
    line = f"{name:9s} out {nout:5d} in {kin:4d}:"
    for v in variants:
        L.set_tuning(f32_split=v)
        t1 = timeit(lambda: gemm(dY, W, dX, R, kin, nout, 0, 1))
        t2 = timeit(lambda: gemm(dY, X, dW, nout, kin, R, 1, 1, acc=1, splitk=sk))
        fl = 2.0 * R * nout * kin
        line += f"  [{v}] dgrad {t1:6.1f} us {fl / t1 * 1e-6:5.1f} TF  wgrad(sk {sk}) {t2:6.1f} us {fl / t2 * 1e-6:5.1f} TF"
    print(line, flush=True)
L.set_tuning(f32_split=1)
