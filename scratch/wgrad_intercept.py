"""Fixed cost of the grouped weight-gradient launch: time against the row count (decoder-layer shapes, 4 K ranges)."""
import sys, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/scratch"]
import torch
from wgrad_group_bench import group, L, lib, timeit
d, ff = 512, 2048
shapes = [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]
s = L.stream_ptr()
for rows in (1024, 2048, 4096, 8192, 16640, 33280):
    a, keep = group(rows, shapes, 4)
    line = f"rows {rows:6d}:"
    for name, fl in (("slabs", 0), ("atomics", 8), ("no-epilogue", 16), ("empty", 16 | 7)):
        a.flags = fl
        line += f"  {name} {timeit(lambda: lib.ortk_wgrad_group(C.byref(a), s)):7.1f} us"
    print(line, flush=True)
