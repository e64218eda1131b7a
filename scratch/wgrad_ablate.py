"""Ablations of the grouped weight-gradient kernel (lock-step loop): no MFMA, no DMA."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/scratch"]
import torch
from wgrad_group_bench import group, L, lib, timeit
d, ff = 512, 2048
rows, shapes = 16640, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]
a, keep = group(rows, shapes, 4)
s = L.stream_ptr()
for name, fl in (("pingpong", 0), ("lockstep", 1), ("lockstep no-mfma", 3), ("lockstep no-dma", 5), ("lockstep neither", 7)):
    a.flags = fl
    print(name, f"{timeit(lambda: lib.ortk_wgrad_group(C.byref(a), s)):.1f} us", flush=True)
