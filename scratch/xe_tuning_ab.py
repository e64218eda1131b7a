"""A/B of the executor's scheduling switches on the XE step of bench.py (BASELINE configs[1]: 256 images x 5 captions, bf16), same box,
interleaved: side stream for the weight gradients on / off  x  rows-stationary chains forward only / + encoder backward / + decoder
backward (ortk_tuning.side_stream, .row_chain)."""
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
combos = [dict(side_stream=s, row_chain=r) for s in (1, 0) for r in (1, 2, 3)] + [dict(side_stream=1, row_chain=1, chain_wide=0)]
res = {i: [] for i in range(len(combos))}
for i, kw in enumerate(combos):
    set_tuning(**kw)
    for _ in range(5): tr.xe_step(batch)
for rep in range(4):
    for i, kw in enumerate(combos):
        set_tuning(**kw)
        for _ in range(2): tr.xe_step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr.xe_step(batch)
        torch.cuda.synchronize(); res[i].append((time.perf_counter() - t0) * 50)
for i, kw in enumerate(combos):
    print(kw, [round(x, 3) for x in res[i]])
lib.ortk_set_tuning(C.byref(base))
