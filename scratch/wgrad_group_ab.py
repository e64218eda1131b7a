"""A/B of the grouped weight gradients inside the XE step of bench.py (BASELINE configs[1]: 256 images x 5 captions, bf16), one box,
interleaved: ortk_tuning.wgrad_group (bit mask: 1 layers, 2 generator, 4 memory K|V, 8 partial tiles in memory instead of atomics) x
.wgrad_group_splitk."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
L = pkg._lib
lib = L.lib()
dev = torch.device("cuda", 0)
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="bf16").to(dev).train()
batch = bench.synth_batch(256, 36, 2048, 10001, 5, 18, 1000, dev)
tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000)
base = L.Tuning(); lib.ortk_get_tuning(C.byref(base))
def set_tuning(**kw):
    t = L.Tuning.from_buffer_copy(base)
    for k, v in kw.items(): setattr(t, k, v)
    assert lib.ortk_set_tuning(C.byref(t)) == 0
combos = [dict(wgrad_group=0)] + [dict(wgrad_group=g, wgrad_group_wgs=t, wgrad_group_tail=tl) for g in (3, 7) for t in (80, 96, 112, 128, 144, 176) for tl in (0, 1)]
if len(sys.argv) > 1:      # field=value,field=value ...
    combos = [dict(wgrad_group=0)] + [{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")} for a in sys.argv[1:]]
res = {i: [] for i in range(len(combos))}
for i, kw in enumerate(combos):
    set_tuning(**kw)
    for _ in range(4): tr.xe_step(batch)
for rep in range(3):
    for i, kw in enumerate(combos):
        set_tuning(**kw)
        for _ in range(2): tr.xe_step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr.xe_step(batch)
        torch.cuda.synchronize(); res[i].append((time.perf_counter() - t0) * 50)
for i, kw in enumerate(combos):
    print(kw, [round(x, 3) for x in res[i]], flush=True)
lib.ortk_set_tuning(C.byref(base))
