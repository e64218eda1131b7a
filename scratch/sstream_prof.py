"""One executor, a few 1 024-image beam-5 decodes on 95 %-pruned weights: the command the rocprofv3 passes of the stack kernel run
(python scratch/sstream_prof.py stack|sparse_stream|sparse_stream_rb20 [n])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config

ex = sys.argv[1] if len(sys.argv) > 1 else "sparse_stream"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16")
with torch.no_grad():
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            p.mul_((torch.rand_like(p) < 0.05).float())
m = m.to(dev).eval()
b = Bn.synth_batch(1024, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
with torch.no_grad():
    for _ in range(n):
        m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt={"beam_size": 5, "executor": ex}, mode="sample")
torch.cuda.synchronize()
print("done", ex)
