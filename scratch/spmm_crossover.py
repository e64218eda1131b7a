"""Where do the sparse products pay?  ortk_spmm (ELL16: the default mixed-precision format; GU16 beside it) against this library's dense
bf16 GEMM on the same zero-filled weights, per projection shape of the path (forward blocks (N, K) and the transposed blocks of the
data gradients), per row count (decode 5 120, encoder 9 216, valid-position decoder 16 640, padded decoder 21 760) and per sparsity
(95 / 97.5 / 98.8 / 99.5 %).  Interleaved rounds in one process.  Writes the table `sparse.py: CROSSOVER` is taken from:
    python scratch/spmm_crossover.py > profiles/r04_spmm_crossover.txt"""
import ctypes as C, json, sys, torch
sys.path.insert(0, "/root/repo")
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.sparse import SparsePlan, capacity_for
L = P._lib; lib = L.lib()


def t_us(fns, n=20, rounds=3):
    best = [1e9] * len(fns)
    for f in fns:
        for _ in range(3): f()
    for _ in range(rounds):
        for i, f in enumerate(fns):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): f()
            b.record(); torch.cuda.synchronize()
            best[i] = min(best[i], a.elapsed_time(b) / n * 1e3)
    return best


# (N outputs, K inputs): forward blocks, then the transposed blocks the data gradients multiply by
NK = [(512, 2048), (512, 512), (1536, 512), (2048, 512), (6144, 512), (10112, 512), (512, 1536), (512, 6144)]
rows = {(512, 2048): (9216, 16640), (512, 512): (5120, 9216, 16640), (1536, 512): (5120, 9216, 16640), (2048, 512): (5120, 9216, 16640),
        (6144, 512): (9216,), (10112, 512): (5120, 16640), (512, 1536): (9216, 16640), (512, 6144): (9216,)}
table = {}
for sp in (0.95, 0.975, 0.988, 0.995):
    for (N, K) in NK:
        for M in rows[(N, K)]:
            W = torch.randn(N, K, device="cuda") * (torch.rand(N, K, device="cuda") >= sp).float()
            W16 = W.bfloat16()
            pieces = [(k0, min(2048, K - k0)) for k0 in range(0, K, 2048)]       # ELL: at most 2 048 inputs per launch
            ell = SparsePlan([dict(offset=k0, N=N, K=kw, ld=K, capacity=capacity_for(N, kw, 0.06)) for k0, kw in pieces], L.SP_ELL16, "cuda")
            ell.build(W16); ell.check_overflow()
            X = torch.randn(M, K, device="cuda").bfloat16()
            Y = torch.empty(M, N, device="cuda", dtype=torch.float32 if len(pieces) > 1 else torch.bfloat16)
            bias = torch.randn(N, device="cuda")
            sas = []
            for i, (k0, kw) in enumerate(pieces):
                sa = L.SpmmArgs(); sa.X, sa.Y, sa.ldx, sa.ldy, sa.M = X.data_ptr() + 2 * k0, Y.data_ptr(), K, N, M
                sa.x_dtype, sa.y_dtype = 1, (0 if len(pieces) > 1 else 1)
                if i == 0: sa.bias = bias.data_ptr()
                else: sa.resid, sa.ldr = Y.data_ptr(), N
                sas.append(sa)

            def f_ell():
                for i, sa in enumerate(sas): lib.ortk_spmm(ell.ref(), i, C.byref(sa), L.stream_ptr())
            a = L.GemmArgs(); a.A, a.B, a.C = L.ptr(X), L.ptr(W16), L.ptr(Y); a.lda, a.ldb, a.ldc = K, K, N
            a.M, a.N, a.K, a.precision, a.a_dtype, a.b_dtype, a.c_dtype = M, N, K, 1, 1, 1, (0 if len(pieces) > 1 else 1); a.bias = L.ptr(bias)
            f_d = lambda: lib.ortk_gemm(C.byref(a), L.stream_ptr())
            f_build = lambda: ell.build(W16)
            def f_ell_plain():           # (K > 512: the form with the output tile beside the planes, one workgroup per CU)
                L.set_tuning(spmm_alias=0)
                f_ell()
                L.set_tuning(spmm_alias=1)
            te, td, tb, tp = t_us([f_ell, f_d, f_build, f_ell_plain] if K > 512 else [f_ell, f_d, f_build])[:4] + ([None] if K <= 512 else [])
            table[f"{sp}/{N}x{K}/{M}"] = (round(te, 1), round(td, 1), round(tb, 1))
            print(f"sp={sp:5.3f} N={N:6d} K={K:5d} M={M:6d}: ell16 {te:7.1f} us  dense {td:7.1f} us  ell16/dense {te / td:5.2f}  build {tb:6.1f} us" +
                  (f"  (tile beside the planes {tp:7.1f} us)" if tp is not None else ""), flush=True)
print("JSON " + json.dumps(table))
