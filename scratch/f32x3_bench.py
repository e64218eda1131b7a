"""fp32 GEMM of the forward layout: native fp32 MFMA kernel (f32_split = 0) against the three-way bf16 split kernel
(gemm_f32x3_kernel / gemm_f32x3p_kernel; f32_split = 1 automatic, 2..7 fix the kernel instance), on the shapes of the fp32 parity decode (1 024 images x 5 beams, 36 regions).

Prints, per shape and variant: us per launch, fp32-equivalent TF/s, and the error against a float64 product of the same fp32
operands: max |err| / max |ref| and rms err / rms ref.   python scratch/f32x3_bench.py [quick]
"""
import ctypes as C
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sparse_image_captioning_amd as pkg   # noqa: E402
from sparse_image_captioning_amd import _lib as L   # noqa: E402


def gemm(A, B, Cout, bias=None, relu=0, resid=None):
    a = L.GemmArgs()
    M, K = A.shape; N = B.shape[0]
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    a.lda, a.ldb, a.ldc = A.stride(0), B.stride(0), Cout.stride(0)
    a.M, a.N, a.K, a.transA, a.transB, a.precision = M, N, K, 0, 0, 0
    if bias is not None: a.bias = bias.data_ptr()
    if resid is not None: a.resid = resid.data_ptr(); a.ldr = resid.stride(0)
    a.relu = relu
    L.check(L.lib().ortk_gemm(C.byref(a), L.stream_ptr()), "ortk_gemm")


def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    shapes = [("d.qkv", 5120, 1536, 512), ("d.wo/cq/co", 5120, 512, 512), ("d.w1", 5120, 2048, 512), ("d.w2", 5120, 512, 2048),
              ("gen", 5120, 10112, 512), ("e.qkv", 36864, 1536, 512), ("e.wo", 36864, 512, 512), ("e.w1", 36864, 2048, 512),
              ("e.w2", 36864, 512, 2048), ("att", 36864, 512, 2048), ("ragged", 5000, 1000, 544)]
    if quick: shapes = shapes[:4]
    g = torch.Generator().manual_seed(1)
    for name, M, N, K in shapes:
        A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * 0.05).cuda()
        bias = torch.randn(N, generator=g).cuda()
        sub = slice(0, min(M, 1024))
        ref = A[sub].double() @ B.double().t() + bias.double()
        line = f"{name:11s} M {M:6d} N {N:6d} K {K:5d}:"
        outs = {}
        for v in (0, 1, 2, 3, 4, 5, 6, 7):
            L.set_tuning(f32_split=v)
            Cout = torch.full((M, N), float("nan"), device="cuda")
            us = timeit(lambda: gemm(A, B, Cout, bias=bias))
            err = (Cout[sub].double() - ref)
            outs[v] = Cout
            line += f"  [{v}] {us:6.1f} us {2.0 * M * N * K / us * 1e-6:5.1f} TF"
            if v in (0, 1):
                line += f" (max {err.abs().max().item() / ref.abs().max().item():.1e} rms {err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item():.1e})"
        assert not torch.isnan(outs[1]).any()
        for v in (2, 3, 4, 5, 6, 7):
            assert torch.equal(outs[1], outs[v]), f"tile shapes differ ({v})"      # same k order, same partial-product order
        print(line, flush=True)
    L.set_tuning(f32_split=1)


if __name__ == "__main__":
    main()
