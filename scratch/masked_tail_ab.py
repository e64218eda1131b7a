"""A/B on one box: the supermask XE step of bench.py (configs[2], masked dense products) with the element-wise tail of the step (mask
backward, clip + Adam on the weights, clip + Adam on the mask logits) as ONE pass over the arena (ortk_masked_adam_step,
NativeTrainer.fused_masked_tail) or as the four launches of rounds 2-4.  (Round 5 also measured the tail's decoder half on a second
stream beside the encoder half of the backward: 11.91 vs 11.92 ms with the one-pass tail, 12.27 vs 12.33 with four launches — dropped.)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
dev = torch.device("cuda", 0)
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer_prune")(config, precision="bf16")
with torch.no_grad():
    for _, m in model.all_pruning_masks():
        m.copy_(torch.where(torch.rand_like(m) < 0.05, torch.full_like(m, 6.0), torch.full_like(m, -6.0)))
model = model.to(dev).train()
batch = bench.synth_batch(256, 36, 2048, 10001, 5, 18, 1000, dev)
tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000, sparsity_target=0.95, max_train_step=100000)
res = {True: [], False: []}
for f in (True, False):
    tr.fused_masked_tail = f
    for _ in range(5): tr.xe_step(batch)
for rep in range(4):
    for f in (True, False):
        tr.fused_masked_tail = f
        for _ in range(2): tr.xe_step(batch)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): tr.xe_step(batch)
        torch.cuda.synchronize(); res[f].append(round((time.perf_counter() - t0) * 50, 3))
print("one pass    ", res[True]); print("four launches", res[False])
