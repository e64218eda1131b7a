"""Bench-size train-mode rollout vs teacher forcing, per ROW: is a row with large |log-prob difference| an outlier of bf16 noise or a
row whose masks differ between the column-split rollout kernel and the teacher-forced pass?  Three comparisons on the same images:
eval-mode rollout vs eval TF, train-mode rollout (split kernel) vs train TF, train-mode rollout (unfused executor) vs train TF."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import common as C, helpers as H
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.utils.config import Config
from sparse_image_captioning_amd.training import NativeTrainer
state = H.torch_state(H.dense_param_shapes(C.FULL_CFG), C.G2_SEED, C.G1_GEN_SCALE, C.G1_EOS_BIAS)
B, ns = 256, 5
m = P.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=1)
m.load_state_dict(state, strict=False); m = m.cuda()
b = {k: v.cuda() for k, v in H.torch_batch(C.make_inputs(seed=73, n_img=B, n_reg=36, feat=2048, vocab=10001, spi=1, ragged=True)).items() if k not in ("seqs", "masks")}
kw = dict(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], mode="sample")
drop_seed, gseed = 0x1357ACE, 4242

def tf_lp(rows, train, seed):
    tf_in = torch.cat([rows.new_full((rows.size(0), 1), C.BOS), rows], 1)
    batch = m._make_batch(*m._prepare(b["att_feats"], b["boxes"], b["att_masks"]), tf_in, rollouts=True)
    with torch.no_grad():
        logp, _ = m._run_forward(batch, train, seed, want_logp=True, cache_ws=False)
    return logp[..., :m.vocab_size].gather(2, rows.unsqueeze(2)).squeeze(2)

def report(name, seq, lp, train):
    rows = seq.reshape(-1, seq.size(-1)); roll = lp.reshape(-1, lp.size(-1))
    err = (tf_lp(rows, train, drop_seed if train else 0) - roll).abs() * (rows != 0)
    per_row = err.sum(1) / (rows != 0).sum(1).clamp(min=1)
    top = torch.topk(per_row, 6)
    print(f"{name}: mean {err[rows != 0].mean():.5f} max {err.max():.4f} frac>0.05 {(err > 0.05).float().sum() / (rows != 0).sum():.5f}")
    print("   rows with the largest MEAN error:", [(int(i), round(float(v), 4), round(float(err[i].max()), 3)) for v, i in zip(top.values, top.indices)])
    return per_row

with torch.no_grad():
    m.eval()
    s, lp = m(**kw, opt={"num_random_sample": ns, "beam_size": 0, "seed": gseed, "executor": "stack_split"})
    report("eval  split", s, lp, False)
    s, lp = m(**kw, opt={"num_random_sample": ns, "beam_size": 0, "seed": gseed, "executor": "unfused"})
    report("eval  unfused", s, lp, False)
    m.train()
    base = {"num_random_sample": ns, "beam_size": 0, "seed": gseed, "train_mode": True, "drop_seed": drop_seed}
    s, lp = m(**kw, opt=dict(base, executor="stack_split"))
    pr = report("train split", s, lp, True)
    s2, lp2 = m(**kw, opt=dict(base, executor="unfused"))
    pr2 = report("train unfused", s2, lp2, True)
    # with the greedy rows riding along (1 536 rows: the scst_step form)
    tr = NativeTrainer(m, noamopt_factor=0.0, noamopt_warmup=10)
    mem = tr.encode_for_update(b, B * ns, train=True, seed=drop_seed)
    s3, lp3 = m(**kw, opt=dict(base, with_greedy=True, memory=mem))
    report("train split + greedy rows + shared memory", s3[:, 1:].contiguous(), lp3[:, 1:].contiguous(), True)
    bad = int(torch.argmax(pr))
    print("worst row of the split rollout:", bad, "tokens", s.reshape(-1, s.size(-1))[bad].tolist())
    print("  same row, unfused executor:   tokens", s2.reshape(-1, s2.size(-1))[bad].tolist(), "mean err", float(pr2[bad]))
