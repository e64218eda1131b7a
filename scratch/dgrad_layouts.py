"""dgrad shapes: k-major weight operand (today: register-staged kernel) vs the same product with a pre-transposed weight (forward layout, LDS-DMA kernels)."""
import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def run(M,N,K,tb,reps=20):
    A=torch.randn(M,K,device="cuda").bfloat16(); B=torch.randn((K,N) if tb else (N,K),device="cuda").bfloat16(); Cc=torch.zeros(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=K,B.stride(0),N
    a.M,a.N,a.K,a.transB,a.precision=M,N,K,tb,1; a.a_dtype=1; a.b_dtype=1; a.c_dtype=0
    for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a),L.stream_ptr()),"g")
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
Me,Md=9216,21760
shapes=[("e.qkv",Me,512,1536,6),("e.wo",Me,512,512,6),("e.w1",Me,512,2048,6),("e.w2",Me,2048,512,6),("ckv",Me,512,6144,1),
        ("d.qkv",Md,512,1536,6),("d.wo",Md,512,512,18),("d.w1",Md,512,2048,6),("d.w2",Md,2048,512,6),("gen",Md,512,10240,1)]
ta=tb=0
for name,M,N,K,cnt in shapes:
    t1=run(M,N,K,1); t0=run(M,N,K,0); ta+=t1*cnt; tb+=t0*cnt
    print(f"{name:6s} M{M:6d} N{N:5d} K{K:6d} x{cnt:2d}: k-major W {t1:7.1f} us   W^T (fwd layout) {t0:7.1f} us")
print("per step: k-major", ta/1e3, "ms; transposed", tb/1e3, "ms")
