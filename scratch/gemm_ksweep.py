import sys
sys.path[:0]=["/root/repo"]
exec(open("/root/repo/scratch/gemm_shapes.py").read().split("Me,Md=")[0])
for N in (512, 2048):
    for cdt in (0, 1):
        for K in (64, 128, 256, 512, 1024, 2048):
            t = run(21760, N, K, 0, 0, cdt=cdt)
            print(f"M21760 N{N} K{K:5d} cdt{cdt}: {t:7.1f} us  {2*21760*N*K/t/1e6:6.0f} TF  out {21760*N*(2 if cdt else 4)/t/1e6:5.2f} TB/s", flush=True)
