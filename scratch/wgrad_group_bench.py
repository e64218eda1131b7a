"""ortk_wgrad_group (one grouped launch per layer) against the per-projection weight-gradient launches it replaces:
correctness against torch fp32 on the same bf16 operands, then us per layer for both, alone on the chip.
Usage: python scratch/wgrad_group_bench.py [splitk ...]"""
import sys, ctypes as C, time
sys.path[:0] = ["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L = P._lib
lib = L.lib()
torch.manual_seed(0)


def group(rows, shapes, splitk=0, ldpad=0):
    """shapes: [(Nout, Kin)]; returns (args, keep, tensors)"""
    a = L.WgradGroupArgs(); a.n = len(shapes); a.rows = rows; a.splitk = splitk
    keep = []
    for i, (n, k) in enumerate(shapes):
        dY = (torch.randn(rows, n + ldpad, device="cuda") * 0.5).bfloat16()
        X = torch.randn(rows, k + ldpad, device="cuda").bfloat16()
        dW = torch.zeros(n, k, device="cuda"); db = torch.zeros(n, device="cuda")
        it = a.item[i]
        it.dY, it.lddy, it.X, it.ldx, it.dW, it.lddw, it.db, it.Nout, it.Kin = dY.data_ptr(), n + ldpad, X.data_ptr(), k + ldpad, dW.data_ptr(), k, db.data_ptr(), n, k
        keep.append((dY, X, dW, db, n, k))
    nb = lib.ortk_wgrad_group_workspace_bytes(C.byref(a))
    if nb:
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        a.ws, a.ws_bytes = ws.data_ptr(), nb
        keep.append(ws)
    return a, keep


def single_args(keep, rows, wgs=384):
    out = []
    for dY, X, dW, db, n, k in [k_ for k_ in keep if isinstance(k_, tuple)]:
        g = L.GemmArgs()
        g.A, g.B, g.C = dY.data_ptr(), X.data_ptr(), dW.data_ptr()
        g.lda, g.ldb, g.ldc = dY.stride(0), X.stride(0), k
        g.M, g.N, g.K, g.transA, g.transB, g.precision = n, k, rows, 1, 1, 1
        g.accumulate = 1; g.a_dtype = 1; g.b_dtype = 1; g.colsum = db.data_ptr()
        tiles = ((n + 127) // 128) * ((k + 127) // 128)
        sk = (3 if rows >= 16384 else 1) if tiles >= 256 else (wgs + tiles // 2) // tiles
        g.splitk = max(1, min(sk, max(1, rows // 512)))
        out.append(g)
    return out


def check(rows, shapes, splitk=0, ldpad=0, flags=0):
    a, keep = group(rows, shapes, splitk, ldpad)
    a.flags = flags
    items = [k_ for k_ in keep if isinstance(k_, tuple)]
    worst = 0.0
    for rep in range(3):          # the same workspace again, on new operands
        for (dY, X, dW, db, _, _) in items:
            if rep:
                dY.copy_((torch.randn_like(dY, dtype=torch.float32) * 0.5).bfloat16()); X.copy_(torch.randn_like(X, dtype=torch.float32).bfloat16())
            dW.fill_(1.0); db.fill_(2.0)
        L.check(lib.ortk_wgrad_group(C.byref(a), L.stream_ptr()), "wgrad_group")
        torch.cuda.synchronize()
        for dY, X, dW, db, n, k in items:
            ref = dY[:, :n].float().t() @ X[:, :k].float() + 1.0
            refb = dY[:, :n].float().sum(0) + 2.0
            e = ((dW - ref).abs().max() / ref.abs().max()).item(); eb = ((db - refb).abs().max() / refb.abs().max()).item()
            worst = max(worst, e, eb)
    print(f"check rows {rows} shapes {shapes} splitk {splitk} ldpad {ldpad} flags {flags}: max rel err {worst:.2e}", flush=True)
    assert worst < 2e-5, worst


def timeit(f, n=30):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


if __name__ == "__main__":
    P._lib.require_gpu()
    for fl in (0, 1, 8):
        check(1000, [(512, 512)], 0, flags=fl)
        check(1000 + 17, [(512, 512), (1536, 512), (512, 2048), (2048, 512)], 0, flags=fl)
        check(333, [(264, 520), (8, 8), (128, 1000)], 0, ldpad=8, flags=fl)
        check(4097, [(10112, 512)], 3, flags=fl)
        check(96, [(512, 512)], 0, flags=fl)
        check(31, [(512, 256)], 0, flags=fl)
        check(32, [(512, 256)], 0, flags=fl)
        check(64 + 5, [(256, 256)], 0, flags=fl)
        check(16640, [(512, 512), (1536, 512)], 4, flags=fl)
    d, ff = 512, 2048
    layers = {
        "decoder layer (16640 rows)": (16640, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]),
        "encoder layer (9216 rows)": (9216, [(d, ff), (ff, d), (d, d), (3 * d, d)]),
        "generator (16640 rows)": (16640, [(10112, d)]),
        "memory K|V (9216 rows)": (9216, [(6 * 2 * d, d)]),
        "decoder layer, padded layout (21760 rows)": (21760, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]),
    }
    sks = [int(x) for x in sys.argv[1:]] or [0]
    s = L.stream_ptr()
    for name, (rows, shapes) in layers.items():
        a, keep = group(rows, shapes)
        singles = single_args(keep, rows)
        fl = sum(2.0 * rows * n * k for n, k in shapes)
        t_old = timeit(lambda: [lib.ortk_gemm(C.byref(g), s) for g in singles])
        line = f"{name}: per-projection launches {t_old:7.1f} us ({fl / t_old / 1e6:5.0f} TF/s)"
        for sk in sks:
            for flg in (0, 1, 8):
                a.splitk = sk; a.flags = flg
                nb = lib.ortk_wgrad_group_workspace_bytes(C.byref(a))
                ws = torch.empty(max(nb, 256), dtype=torch.uint8, device="cuda"); a.ws, a.ws_bytes = ws.data_ptr(), nb
                t_new = timeit(lambda: lib.ortk_wgrad_group(C.byref(a), s))
                line += f" | grouped splitk {sk}{' lockstep' if flg == 1 else ' atomics' if flg == 8 else ''}: {t_new:7.1f} us ({fl / t_new / 1e6:5.0f} TF/s)"
        print(line, flush=True)
