import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def run(M,N,K,ta,tb,adt,bdt,reps=3):
    dt={0:torch.float32,1:torch.bfloat16}
    A=torch.randn((K,M) if ta else (M,K),device="cuda").to(dt[adt]); B=torch.randn((K,N) if tb else (N,K),device="cuda").to(dt[bdt]); Cc=torch.zeros(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=A.stride(0),B.stride(0),N
    a.M,a.N,a.K,a.transA,a.transB,a.precision=M,N,K,ta,tb,1; a.a_dtype=adt; a.b_dtype=bdt
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    torch.cuda.synchronize()
run(21760,2048,512,0,0,1,1)
run(21760,512,512,0,0,1,1)
run(4096,4096,4096,0,0,1,1)
run(21760,2048,512,0,0,0,0)
