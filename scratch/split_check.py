"""Column-split decoder stack kernel (executor "stack_split") vs the plain stack kernel: tokens / log-probs and the time of a decode
(python scratch/split_check.py [images])."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev).eval()
b = Bn.synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
m32 = pkg.get_model("relation_transformer")(cfg, precision="fp32").to(dev).eval()
m32.load_state_dict(m.state_dict())


def tf_err(seq, lp):
    """teacher-forced fp32 log-probs of the decoded tokens vs the log-probs the decode reported (first hypothesis of every image)"""
    n = min(seq.size(0), 64)
    rows = seq[:n, 0]
    with torch.no_grad():
        tf_in = torch.cat([rows.new_full((rows.size(0), 1), 2), rows], 1)
        ref = m32(att_feats=b["att_feats"][:n], boxes=b["boxes"][:n], seqs=tf_in, att_masks=b["att_masks"][:n]).gather(2, rows.unsqueeze(2)).squeeze(2)
    e = (lp[:n, 0] - ref)[rows != 0].abs()
    return f"max {e.max().item():.4f} mean {e.mean().item():.4f}"



DBG = int(os.environ.get("DBG", "0"))


def run(ex, opt, n=5):
    o = dict(opt, executor=ex, stack_debug=DBG)
    with torch.no_grad():
        seq, lp = m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=o, mode="sample")
        torch.cuda.synchronize()
    return seq, lp, (time.perf_counter() - t0) / n * 1e3


opts = ({"beam_size": 1}, {"beam_size": 5}, {"num_random_sample": 5, "beam_size": 0, "with_greedy": True})
if len(sys.argv) > 2:
    opts = opts[:int(sys.argv[2])]
for opt in opts:
    sd, ld, td = run("stack", opt)
    ss, ls, ts = run(os.environ.get("EX2", "stack_split"), opt)
    if os.environ.get("G2"):
        os.environ["ORTK_EXP_TPG"] = os.environ["G2"]
        su, lu, tu = run("stack_split", opt)
        del os.environ["ORTK_EXP_TPG"]
    else:
        su, lu, tu = run("unfused", opt)
    same = sd == ss
    d = (ld - ls)[same].abs()
    print(f"{opt}: stack {td:.2f} ms, split {ts:.2f} ms, unfused {tu:.2f} ms; tokens equal (split vs stack) {same.float().mean().item():.4f}, "
          f"(unfused vs stack) {(sd == su).float().mean().item():.4f}; rows equal {(sd.flatten(0, -2) == ss.flatten(0, -2)).all(-1).float().mean().item():.4f}, "
          f"|dlogp| max {d.max().item():.4g} mean {d.mean().item():.3g}", flush=True)
    print("   error against fp32 teacher forcing: stack", tf_err(sd, ld), "| split", tf_err(ss, ls), "| unfused", tf_err(su, lu), flush=True)
    a2, b2 = sd.flatten(0, -2), ss.flatten(0, -2)
    print("   per position token agreement:", [round((a2[:, t] == b2[:, t]).float().mean().item(), 3) for t in range(min(6, a2.size(1)))],
          " |dlogp| at t=0:", (ld.flatten(0, -2)[:, 0] - ls.flatten(0, -2)[:, 0]).abs().max().item(), flush=True)
