import sys, ctypes as C
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
exec(open("/root/repo/scratch/gemm_bench.py").read().split("for (adt,bdt)")[0])
for (M,N) in ((512,512),(1536,512),(2048,512),(512,2048)):
    for sk in (4,8,12,16,24,32,48):
        run(M,N,21760,1,1,1,1,1,acc=1,splitk=sk)
for sk in (1,2,3,4): run(10112,512,21760,1,1,1,1,1,acc=1,splitk=sk)
for sk in (2,3,4,6): run(6144,512,9216,1,1,1,1,1,acc=1,splitk=sk)
