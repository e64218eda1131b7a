"""Run a few launches of the three GEMM layouts at the decoder QKV shape (for rocprofv3 --pmc passes)."""
import sys
sys.path[:0]=["/root/repo"]
exec(open("/root/repo/scratch/gemm_shapes.py").read().split("Me,Md=")[0])
M,N,K=21760,1536,512
print("fwd", run(M,N,K,0,0,cdt=0,reps=5))
print("fwd512", run(M,512,512,0,0,cdt=0,reps=5))
print("dgrad", run(M,K,N,0,1,cdt=0,reps=5))
print("wgrad", run(N,K,M,1,1,acc=1,splitk=8,reps=5))
