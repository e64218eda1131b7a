"""Sparse weight stream of the decoder stack kernel vs its dense stream on the same 95 %-pruned weights: tokens / log-probs,
and the time of a 1 024-image beam-5 decode with each (python scratch/sstream_check.py [images] [sparsity])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sp = float(sys.argv[2]) if len(sys.argv) > 2 else 0.95
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16")
with torch.no_grad():
    for _, p in m.named_parameters():
        if p.dim() >= 2:
            p.mul_((torch.rand_like(p) < 1 - sp).float())
m = m.to(dev).eval()
b = Bn.synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)


def run(ex, opt=None, n=5, dbg=0):
    o = dict(opt or {"beam_size": 5}, executor=ex, stack_debug=dbg)
    with torch.no_grad():
        seq, lp = m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
        torch.cuda.synchronize()
    return seq, lp, (time.perf_counter() - t0) / n * 1e3


for opt, ex in (({"beam_size": 5}, "sparse_stream"), ({"beam_size": 1}, "sparse_stream")):
    sd, ld, td = run("stack", opt)
    ss, ls, ts = run(ex, opt)
    same = sd == ss
    d = (ld - ls)[same].abs()
    print(f"{opt} {ex}: dense stream {td:.2f} ms, sparse stream {ts:.2f} ms; tokens equal {same.float().mean().item():.4f}, first-beam rows equal "
          f"{(sd[:, 0] == ss[:, 0]).all(-1).float().mean().item():.4f}, |dlogp| max {d.max().item():.4g} mean {d.mean().item():.3g}", flush=True)
if "--phases" in sys.argv:
    for ex in ("sparse_stream",):
        for dbg in (0, 1, 2, 4, 7):
            _, _, t = run(ex, {"beam_size": 5}, 3, dbg)
            print(f"{ex} debug {dbg}: {t:.2f} ms", flush=True)
