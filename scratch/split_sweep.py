"""Decode time by executor over batch sizes (beam 5): python scratch/split_sweep.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev).eval()
for B, beam in ((20, 5), (50, 5), (100, 5), (200, 5), (320, 5), (400, 5), (512, 5), (700, 5), (1024, 5), (256, 1), (1024, 1), (2048, 1)):
    b = Bn.synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
    res = []
    for ex in ("unfused", "stack", "stack_split"):
        o = {"beam_size": beam, "executor": ex}
        with torch.no_grad():
            m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 4 * 1e3)
    print(f"images {B:5d} beam {beam} rows {B * beam:5d}: unfused {res[0]:6.2f}  stack {res[1]:6.2f}  split {res[2]:6.2f} ms", flush=True)
