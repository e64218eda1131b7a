"""Cost of the epilogue options on forward-layout GEMMs of the step (bf16 operands)."""
import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def run(M,N,K,drop,resid,relu,cdt,reps=20):
    A=torch.randn(M,K,device="cuda").bfloat16(); B=torch.randn(N,K,device="cuda").bfloat16()
    Cc=torch.zeros(M,N,device="cuda",dtype=torch.bfloat16 if cdt else torch.float32)
    bias=torch.randn(N,device="cuda"); res=torch.randn(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=K,K,N
    a.M,a.N,a.K,a.precision=M,N,K,1; a.a_dtype=1; a.b_dtype=1; a.c_dtype=cdt; a.bias=bias.data_ptr(); a.relu=relu
    if resid: a.resid=res.data_ptr(); a.ldr=N
    a.drop_p=drop; a.drop_seed=7
    for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a),L.stream_ptr()),"g")
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
for name,M,N,K,resid,relu,cdt in (("d.w1",21760,2048,512,0,1,1),("d.w2",21760,512,2048,1,0,0),("d.wo",21760,512,512,1,0,0),("e.w1",9216,2048,512,0,1,1)):
    t0=run(M,N,K,0.0,resid,relu,cdt); t1=run(M,N,K,0.1,resid,relu,cdt)
    print(f"{name} M{M} N{N} K{K}: no dropout {t0:6.1f} us, dropout 0.1 {t1:6.1f} us")
