"""Time every GEMM shape of one XE step (B=256) in isolation, mixed precision, and print the per-step budget."""
import sys, ctypes as C, io, contextlib
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L=P._lib
dt={0:torch.float32,1:torch.bfloat16}
def run(M,N,K,ta,tb,acc=0,splitk=1,reps=10,cdt=0):
    A=torch.randn((K,M) if ta else (M,K),device="cuda").to(dt[1]); B=torch.randn((K,N) if tb else (N,K),device="cuda").to(dt[1]); Cc=torch.zeros(M,N,device="cuda",dtype=dt[cdt])
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=A.stride(0),B.stride(0),N
    a.M,a.N,a.K,a.transA,a.transB,a.precision=M,N,K,ta,tb,1; a.accumulate=acc; a.splitk=splitk; a.a_dtype=1; a.b_dtype=1; a.c_dtype=cdt
    for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a),L.stream_ptr()),"g")
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps
Me,Md=9216,21760
fwd=[("att",Me,512,2048,1),("e.qkv",Me,1536,512,6),("e.wo",Me,512,512,6),("e.w1",Me,2048,512,6),("e.w2",Me,512,2048,6),("ckv",Me,6144,512,1),
     ("d.qkv",Md,1536,512,6),("d.wo/cq/co",Md,512,512,18),("d.w1",Md,2048,512,6),("d.w2",Md,512,2048,6),("gen",Md,10240,512,1)]
def splitk(M,Nout,Kin):
    tiles=((Nout+127)//128)*((Kin+127)//128)
    sk = (3 if M>=16384 else 1) if tiles>=256 else (384+tiles//2)//tiles
    return max(1,min(sk,max(1,M//512)))
tot={"fwd":0,"dgrad":0,"wgrad":0}
for name,M,N,K,cnt in fwd:
    t=run(M,N,K,0,0); tot["fwd"]+=t*cnt
    print(f"fwd   {name:10s} M{M:6d} N{N:5d} K{K:5d} x{cnt:2d}: {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF  -> {t*cnt/1e3:6.3f} ms", flush=True)
for name,M,N,K,cnt in fwd:
    if name=="att": continue
    t=run(M,K,N,0,1); tot["dgrad"]+=t*cnt
    print(f"dgrad {name:10s} M{M:6d} N{K:5d} K{N:5d} x{cnt:2d}: {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF  -> {t*cnt/1e3:6.3f} ms", flush=True)
for name,M,N,K,cnt in fwd:
    sk=splitk(M,N,K)
    t=run(N,K,M,1,1,acc=1,splitk=sk); tot["wgrad"]+=t*cnt
    print(f"wgrad {name:10s} M{N:6d} N{K:5d} K{M:5d} x{cnt:2d} sk{sk:2d}: {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF  -> {t*cnt/1e3:6.3f} ms", flush=True)
print(tot, sum(tot.values()))
