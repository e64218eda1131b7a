"""One grouped weight-gradient launch shape, a few times (for rocprofv3 --pmc / --kernel-trace).  argv: layer splitk reps"""
import sys, ctypes as C
sys.path[:0] = ["/root/repo", "/root/repo/scratch"]
import torch
import sparse_image_captioning_amd as P
from wgrad_group_bench import group, L, lib
d, ff = 512, 2048
LAY = {"dec": (16640, [(d, ff), (ff, d), (d, d), (d, d), (d, d), (3 * d, d)]), "enc": (9216, [(d, ff), (ff, d), (d, d), (3 * d, d)]),
       "gen": (16640, [(10112, d)]), "ckv": (9216, [(6 * 2 * d, d)])}
name = sys.argv[1] if len(sys.argv) > 1 else "dec"
sk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows, shapes = LAY[name]
a, keep = group(rows, shapes, sk)
for _ in range(reps):
    L.check(lib.ortk_wgrad_group(C.byref(a), L.stream_ptr()), "wgrad_group")
torch.cuda.synchronize()
print("done", name, sk, reps)
