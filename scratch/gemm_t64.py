"""Forward-layout GEMM at decode-sized row counts: run once with ORTK_GEMM_T64=-1 (128^2 / 256^2 kernels) and once with the
default (64^2 tiles on short grids)."""
import sys, ctypes as C, os
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
import os as _os
if "ORTK_GEMM_T64" in _os.environ:      # (the library itself reads no environment: forward the old switch through ortk_set_tuning)
    P._lib.set_tuning(gemm_t64=int(_os.environ["ORTK_GEMM_T64"]))

L=P._lib
def run(M,N,K,reps=20,epi=True):
    A=torch.randn(M,K,device="cuda").bfloat16(); B=torch.randn(N,K,device="cuda").bfloat16(); Cc=torch.zeros(M,N,device="cuda")
    bias=torch.randn(N,device="cuda"); res=torch.randn(M,N,device="cuda")
    a=L.GemmArgs(); a.A,a.B,a.C=A.data_ptr(),B.data_ptr(),Cc.data_ptr(); a.lda,a.ldb,a.ldc=K,K,N
    a.M,a.N,a.K,a.precision=M,N,K,1; a.a_dtype=1; a.b_dtype=1; a.c_dtype=0
    if epi: a.bias=bias.data_ptr(); a.resid=res.data_ptr(); a.ldr=N
    for _ in range(3): L.check(L.lib().ortk_gemm(C.byref(a),L.stream_ptr()),"g")
    ref=(A.float()@B.float().t()+bias+res) if epi else A.float()@B.float().t()
    err=(Cc-ref).abs().max().item()/ref.abs().max().item()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.lib().ortk_gemm(C.byref(a),L.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e3/reps, err
print("T64 =", os.environ.get("ORTK_GEMM_T64","default"))
for M in (256,1024,1536,5120,9216,21760):
    for N,K in ((512,512),(1536,512),(2048,512),(512,2048),(10240,512)):
        t,err=run(M,N,K)
        print(f"M{M:6d} N{N:6d} K{K:5d}: {t:7.1f} us {2*M*N*K/t/1e6:6.0f} TF  tiles128 {M//128*(N//128):5d} err {err:.1e}", flush=True)
