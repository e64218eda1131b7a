import sys, ctypes as C, time
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
def run(M,N,K,ta,tb,prec,acc=0,splitk=1,reps=20,bias=False,resid=False):
    A=torch.randn((K,M) if ta else (M,K),device="cuda"); B=torch.randn((K,N) if tb else (N,K),device="cuda"); Cc=torch.zeros(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=A.stride(0),B.stride(0),N
    a.M,a.N,a.K,a.transA,a.transB,a.precision=M,N,K,ta,tb,prec; a.accumulate=acc; a.splitk=splitk
    if bias: bb=torch.randn(N,device="cuda"); a.bias=bb.data_ptr()
    if resid: rr=torch.randn(M,N,device="cuda"); a.resid=rr.data_ptr(); a.ldr=N
    for _ in range(3): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/reps
    print(f"M{M} N{N} K{K} ta{ta} tb{tb} prec{prec} acc{acc} sk{splitk} b{int(bias)} r{int(resid)}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF")
for prec in (1,0):
    run(21760,2048,512,0,0,prec)
    run(21760,512,512,0,0,prec)
    run(21760,512,512,0,0,prec,bias=True,resid=True)
    run(2048,2048,512,0,0,prec)
    run(4096,4096,4096,0,0,prec)
    run(8192,8192,512,0,0,prec)
    run(21760,512,2048,0,0,prec)
    run(21760,2048,512,0,1,prec)
    run(512,512,21760,1,1,prec,acc=1,splitk=48)
    run(512,512,21760,1,1,prec,acc=1,splitk=8)
    run(2048,512,21760,1,1,prec,acc=1,splitk=12)
    run(4096,4096,4096,1,1,prec)
