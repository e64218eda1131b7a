"""A/B of ortk_tuning switches inside the beam-5 decode of bench.py (1 024 images, mixed precision), one box, interleaved.
argv: field=value,field=value ... (one combination per argument); the first arm is the library's default."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
L = pkg._lib
lib = L.lib()
dev = torch.device("cuda", 0)
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="bf16").to(dev).eval()
batch = bench.synth_batch(1024, 36, 2048, 10001, 5, 18, 1000, dev)
base = L.Tuning(); lib.ortk_get_tuning(C.byref(base))
def set_tuning(**kw):
    t = L.Tuning.from_buffer_copy(base)
    for k, v in kw.items(): setattr(t, k, v)
    assert lib.ortk_set_tuning(C.byref(t)) == 0
combos = [dict()] + [{kv.split("=")[0]: int(kv.split("=")[1]) for kv in a.split(",")} for a in sys.argv[1:]]
def step():
    model(att_feats=batch["att_feats"], boxes=batch["boxes"], att_masks=batch["att_masks"], opt={"beam_size": 5}, mode="sample", att_max_len=batch["att_max_len"])
res = {i: [] for i in range(len(combos))}
for i, kw in enumerate(combos):
    set_tuning(**kw)
    for _ in range(3): step()
for rep in range(3):
    for i, kw in enumerate(combos):
        set_tuning(**kw)
        step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): step()
        torch.cuda.synchronize(); res[i].append((time.perf_counter() - t0) * 100)
for i, kw in enumerate(combos):
    print(kw or "default", [round(x, 3) for x in res[i]], flush=True)
lib.ortk_set_tuning(C.byref(base))
