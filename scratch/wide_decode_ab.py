import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
L = pkg._lib
dev = torch.device("cuda", 0)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
torch.manual_seed(8888)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev).eval()
b = Bn.synth_batch(1024, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
best = {}
for rnd in range(4):
    for cw in (1, 2):
        L.set_tuning(chain_wide=cw)
        o = {"beam_size": 5}
        with torch.no_grad():
            for _ in range(2): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
            torch.cuda.synchronize()
        best[cw] = min(best.get(cw, 1e9), (time.perf_counter() - t0) / 5 * 1e3)
print("decode, encoder chains 48-row form (3 rounds): %.2f ms   76-row form (2 rounds): %.2f ms" % (best[1], best[2]))
