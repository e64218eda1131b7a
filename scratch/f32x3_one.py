"""A few launches of the split fp32 GEMM instances on two shapes (for rocprofv3 --pmc passes):  python scratch/f32x3_one.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from f32x3_bench import gemm, L   # noqa: E402

g = torch.Generator().manual_seed(1)
for M, N, K in ((5120, 1536, 512), (36864, 2048, 512)):
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * 0.05).cuda()
    Cout = torch.empty(M, N, device="cuda")
    for v in (0, 3, 4, 7):
        L.set_tuning(f32_split=v)
        for _ in range(3):
            gemm(A, B, Cout)
        torch.cuda.synchronize()
L.set_tuning(f32_split=1)
