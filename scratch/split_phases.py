"""Time of a decode on the column-split stack kernel with phases skipped (stack_debug bits: 1 self-attention, 2 cross-attention,
4 FFN): python scratch/split_phases.py [images] [beam]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
beam = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev).eval()
b = Bn.synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
exs = sys.argv[3].split(",") if len(sys.argv) > 3 else ("stack_split",)
for ex in exs:
    for dbg in (0, 1, 2, 4, 7):
        o = {"beam_size": beam, "executor": ex, "stack_debug": dbg}
        with torch.no_grad():
            m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize()
        print(f"{ex} images {B} beam {beam} debug {dbg}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
