import sys, ctypes as C, time
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
dt={0:torch.float32,1:torch.bfloat16}
def run(M,N,K,ta,tb,prec,adt=0,bdt=0,acc=0,splitk=1,reps=20,bias=False,resid=False,cdt=0):
    A=torch.randn((K,M) if ta else (M,K),device="cuda").to(dt[adt]); B=torch.randn((K,N) if tb else (N,K),device="cuda").to(dt[bdt]); Cc=torch.zeros(M,N,device="cuda",dtype=dt[cdt])
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=A.stride(0),B.stride(0),N
    a.M,a.N,a.K,a.transA,a.transB,a.precision=M,N,K,ta,tb,prec; a.accumulate=acc; a.splitk=splitk; a.a_dtype=adt; a.b_dtype=bdt; a.c_dtype=cdt
    if bias: bb=torch.randn(N,device="cuda"); a.bias=bb.data_ptr()
    if resid: rr=torch.randn(M,N,device="cuda"); a.resid=rr.data_ptr(); a.ldr=N
    for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a),L.stream_ptr()),"g")
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/reps
    print(f"M{M} N{N} K{K} ta{ta} tb{tb} prec{prec} a{adt} b{bdt} c{cdt} acc{acc} sk{splitk} b{int(bias)} r{int(resid)}: {us:8.1f} us  {2*M*N*K/us/1e6:7.1f} TF", flush=True)
for (adt,bdt) in ((1,1),(0,0)):
    run(21760,2048,512,0,0,1,adt,bdt)
    run(21760,2048,512,0,0,1,adt,bdt,cdt=1)
    run(21760,512,512,0,0,1,adt,bdt)
    run(21760,512,512,0,0,1,adt,bdt,bias=True,resid=True)
    run(4096,4096,4096,0,0,1,adt,bdt)
    run(8192,8192,8192,0,0,1,adt,bdt,reps=5)
    run(21760,512,2048,0,0,1,adt,bdt)
    run(21760,2048,512,0,1,1,adt,bdt)
    run(512,512,21760,1,1,1,adt,bdt,acc=1,splitk=48)
    run(512,512,21760,1,1,1,adt,bdt,acc=1,splitk=8)
    run(2048,512,21760,1,1,1,adt,bdt,acc=1,splitk=12)
    run(4096,4096,4096,1,1,1,adt,bdt)
    run(21760,10112,512,0,0,1,adt,bdt)
