import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_bench.py")).read().split("for (adt,bdt)")[0])
for (M, N, K) in ((16640, 512, 10112), (9216, 512, 6144), (16640, 512, 2048), (9216, 512, 2048), (16640, 512, 512), (16640, 2048, 512), (16640, 1536, 512), (9216, 1536, 512), (9216, 2048, 512), (9216, 512, 512)):
    run(M, N, K, 0, 0, 1, 1, 1)
