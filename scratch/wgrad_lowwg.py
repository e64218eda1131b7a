"""Grouped weight-gradient kernel at the workgroup counts the executor uses (56 / 96 / 80): us per launch and per 32-row stage."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/scratch"]
import torch
from wgrad_group_bench import group, L, lib, timeit
d, ff = 512, 2048
s = L.stream_ptr()
for name, rows, shapes, sk in (("decoder layer", 16640, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)], 1),
                               ("decoder layer", 16640, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)], 2),
                               ("encoder layer", 9216, [(d, ff), (ff, d), (d, d), (3 * d, d)], 2),
                               ("encoder layer", 9216, [(d, ff), (ff, d), (d, d), (3 * d, d)], 1),
                               ("generator", 16640, [(10112, d)], 1)):
    a, keep = group(rows, shapes, sk)
    tiles = sum(((n + 255) // 256) * ((k + 255) // 256) for n, k in shapes)
    for fl, nm in ((0, "ping-pong"), (1, "lock step")):
        a.flags = fl
        t = timeit(lambda: lib.ortk_wgrad_group(C.byref(a), s))
        st = rows / sk / 32
        print(f"{name} rows {rows} splitk {sk} ({tiles * sk} wgs) {nm}: {t:7.1f} us, {t / st * 1e3:6.1f} ns per stage, {sum(2.0 * rows * n * k for n, k in shapes) / t / 1e6:5.0f} TF/s = {sum(2.0 * rows * n * k for n, k in shapes) / t / 1e6 / (tiles * sk) * 256 / 2500 * 100:4.1f} % of the MFMA peak of the units it holds", flush=True)
