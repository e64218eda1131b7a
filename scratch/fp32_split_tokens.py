"""Bench-size check of the fp32 parity decode: 1 024 synthetic images x beam 5 on random-init weights (flat logits: the hardest case for
token agreement), products on the bf16 matrix cores (f32_split = 1) against the fp32 MFMA kernels (f32_split = 0): captions that
differ, and the score gap where they do.    python scratch/fp32_split_tokens.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench   # noqa: E402
import sparse_image_captioning_amd as pkg   # noqa: E402
from sparse_image_captioning_amd.utils.config import ort_config   # noqa: E402

dev = torch.device("cuda:0")
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="fp32").to(dev).eval()
b = bench.synth_batch(1024, 36, config.att_feat_size, config.vocab_size, 5, config.max_seq_length, 1000, dev)
out = {}
for v in (0, 1):
    pkg._lib.set_tuning(f32_split=v)
    seq, lp = model(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt={"beam_size": 5}, mode="sample", att_max_len=b["att_max_len"])
    torch.cuda.synchronize()
    out[v] = (seq.cpu(), lp.cpu())
pkg._lib.set_tuning(f32_split=1)
s0, l0 = out[0]; s1, l1 = out[1]
diff = (s0 != s1).any(-1)
print("images:", s0.shape[0], "captions that differ:", int(diff.sum()))
print("max |log-prob difference| over identical captions:", float((l0 - l1)[~diff].abs().max()))
for i in diff.nonzero().flatten().tolist()[:10]:
    print(" image", i, "scores", float(l0[i].sum()), float(l1[i].sum()), "gap", abs(float(l0[i].sum()) - float(l1[i].sum())))
