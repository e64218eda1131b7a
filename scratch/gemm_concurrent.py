"""Do a dgrad and a wgrad GEMM of the same projection overlap when issued on two streams? (potential of tail filling)"""
import sys, ctypes as C, time
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def mk(M,N,K,ta,tb,acc=0,splitk=1):
    A=torch.randn((K,M) if ta else (M,K),device="cuda").bfloat16(); B=torch.randn((K,N) if tb else (N,K),device="cuda").bfloat16()
    Cc=torch.zeros(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=A.stride(0),B.stride(0),N
    a.M,a.N,a.K,a.transA,a.transB,a.precision=M,N,K,ta,tb,1; a.accumulate=acc; a.splitk=splitk; a.a_dtype=1; a.b_dtype=1
    return a,(A,B,Cc)
lib=L.lib()
for name,(dg,wg) in {"w1": ((21760,512,2048,0,1),(2048,512,21760,1,1,1,6)), "wo": ((21760,512,512,0,1),(512,512,21760,1,1,1,24)), "qkv": ((21760,512,1536,0,1),(1536,512,21760,1,1,1,8))}.items():
    a1,k1=mk(*dg); a2,k2=mk(*wg)
    s1,s2=torch.cuda.Stream(),torch.cuda.Stream()
    def seq(n=20):
        for _ in range(n):
            lib.ortk_gemm(C.byref(a1), s1.cuda_stream); lib.ortk_gemm(C.byref(a2), s1.cuda_stream)
    def par(n=20):
        for _ in range(n):
            lib.ortk_gemm(C.byref(a1), s1.cuda_stream); lib.ortk_gemm(C.byref(a2), s2.cuda_stream)
    for f in (seq,par,seq,par):
        f(3); torch.cuda.synchronize(); t=time.perf_counter(); f(); torch.cuda.synchronize()
        print(name, f.__name__, f"{(time.perf_counter()-t)/20*1e6:.1f} us per pair", flush=True)
