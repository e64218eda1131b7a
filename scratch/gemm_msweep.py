import sys
sys.path[:0]=["/root/repo"]
exec(open("/root/repo/scratch/gemm_shapes.py").read().split("Me,Md=")[0])
for K in (64, 512):
    for M in (128, 1024, 4096, 8192, 16384, 32768, 65536, 131072):
        t = run(M, 512, K, 0, 0, cdt=1)
        nb = (M // 128) * 4
        print(f"M{M:7d} N512 K{K:4d}: {t:7.1f} us  blocks {nb:5d} rounds {nb/512:6.2f}  us/round {t/max(1,nb/512):6.2f}", flush=True)
