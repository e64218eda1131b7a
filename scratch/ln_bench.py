"""ortk_layernorm_bwd at the model width (512), as the XE step calls it (residual gradient in, masked bf16 copy out, dropout 0.1): the
four-column form (ortk_tuning.ln_fuse & 4), the eight-column form, and the eight-column form on a bf16 output gradient."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L = P._lib; lib = L.lib()
def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for rows in (16640, 21760, 9216):
    d = 512
    x = torch.randn(rows, d, device="cuda"); dy = torch.randn(rows, d, device="cuda"); a = torch.randn(d, device="cuda"); b = torch.randn(d, device="cuda")
    dy16 = dy.bfloat16()
    y = torch.empty(rows, d, device="cuda"); st = torch.empty(rows, 2, device="cuda"); dres = torch.randn(rows, d, device="cuda")
    dx = torch.empty_like(x); da = torch.zeros(d, device="cuda"); db = torch.zeros(d, device="cuda"); dz = torch.empty(rows, d, device="cuda", dtype=torch.bfloat16)
    L.check(lib.ortk_layernorm_fwd(L.ptr(x), L.ptr(a), L.ptr(b), L.ptr(y), 0, L.ptr(st), rows, d, 1e-6, L.stream_ptr()), "f")
    def run(g, dt):
        return lambda: lib.ortk_layernorm_bwd_dt(L.ptr(g), dt, L.ptr(x), L.ptr(a), L.ptr(st), L.ptr(dres), L.ptr(dx), L.ptr(da), L.ptr(db), rows, d, 1e-6,
                                                 L.ptr(dz), 1, 0.1, 7, None, L.stream_ptr())
    L.set_tuning(ln_fuse=4); t4 = timeit(run(dy, 0)); dx4, dz4 = dx.clone(), dz.clone()
    L.set_tuning(ln_fuse=0); t8 = timeit(run(dy, 0))
    assert (dx - dx4).abs().max().item() <= 2e-6 * dx4.abs().max().item(), "eight-column form != four-column form"      # (the row sums add in another order)
    t16 = timeit(run(dy16, 1))
    err = (dx - dx4).abs().max().item() / dx4.abs().max().item()
    print(f"rows {rows}: four columns {t4:.1f} us | eight columns {t8:.1f} us | eight columns, bf16 dy {t16:.1f} us (dx rel. diff {err:.1e})", flush=True)
