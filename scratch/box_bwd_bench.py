"""ortk_box_logbias_bwd alone on the chip: us per launch for 1 / 5 / 6 layers at the XE step's size (256 images x 36 regions, 8 heads)."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo"]
import torch
import sparse_image_captioning_amd as P
L = P._lib; lib = L.lib()
B, S, H = 256, 36, 8
g = torch.Generator().manual_seed(0)
xy = torch.rand(B, S, 2, generator=g) * 0.6
boxes = torch.cat([xy, xy + 0.05 + torch.rand(B, S, 2, generator=g) * 0.3], 2).cuda()
dm = (C.c_float * 8)(*[1.0 / (1000.0 ** (k / 8.0)) for k in range(8)])
for nl in (1, 5, 6):
    wg = [torch.randn(H, 64, device="cuda") * 0.1 for _ in range(nl)]; bg = [torch.rand(H, device="cuda") for _ in range(nl)]
    dwg = [torch.zeros(H, 64, device="cuda") for _ in range(nl)]; dbg = [torch.zeros(H, device="cuda") for _ in range(nl)]
    ds = torch.randn(nl, B, H, S, S, device="cuda")
    arr = lambda ts: (C.c_void_p * nl)(*[t.data_ptr() for t in ts])
    a_wg, a_bg, a_dwg, a_dbg = arr(wg), arr(bg), arr(dwg), arr(dbg)
    def run():
        L.check(lib.ortk_box_logbias_bwd(L.ptr(boxes), a_wg, a_bg, dm, L.ptr(ds), a_dwg, a_dbg, nl, B, S, H, L.stream_ptr()), "box bwd")
    run(); run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print(f"{nl} layers: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
