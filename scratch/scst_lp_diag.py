"""Where do the largest |teacher-forced log-prob - rollout log-prob| of a bench-size train-mode SCST rollout sit?  (Noise of two bf16
kernel families, or a mask that differs at some rows / positions?)"""
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
for B in (12, 256):
    m = P.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=1)
    m.load_state_dict(state, strict=False); m = m.cuda()
    ns = 5
    b = {k: v.cuda() for k, v in H.torch_batch(C.make_inputs(seed=73, n_img=B, n_reg=36, feat=2048, vocab=10001, spi=1, ragged=True)).items() if k not in ("seqs", "masks")}
    tr = NativeTrainer(m, noamopt_factor=0.0, noamopt_warmup=10, keep_grads=True)
    m.train(); m._seed_counter = 40
    rw = torch.randn(B * ns).cuda()
    loss, _, seq, greedy = tr.scst_step(b, lambda s_, g_: rw, num_samples=ns)
    rlp = m._last_decode[1].clone()
    rows = seq.reshape(-1, seq.size(-1))
    tf_in = torch.cat([rows.new_full((rows.size(0), 1), C.BOS), rows], 1)
    drop_seed = (torch.initial_seed() * 1000003 + 41) & 0xFFFFFFFFFFFFFFFF or 1
    batch = m._make_batch(*m._prepare(b["att_feats"], b["boxes"], b["att_masks"]), tf_in, rollouts=True)
    with torch.no_grad():
        logp, _ = m._run_forward(batch, True, drop_seed, want_logp=True, cache_ws=False)
    tf = logp[..., :m.vocab_size].gather(2, rows.unsqueeze(2)).squeeze(2)
    roll = rlp[:, 1:].reshape(-1, rlp.size(-1))
    err = (tf - roll).abs() * (rows != 0)
    print(f"B {B}: rows {rows.size(0)} mean len {(rows != 0).sum(1).float().mean():.2f}  err mean {err[rows != 0].mean():.5f} max {err.max():.4f}  frac>0.05 {(err > 0.05).float().sum() / (rows != 0).sum():.5f}")
    print("  by position t:", [round(float(err[:, t].max()), 3) for t in range(err.size(1))])
    print("  mean by position:", [round(float(err[:, t].sum() / max(1, int((rows[:, t] != 0).sum()))), 4) for t in range(err.size(1))])
    top = torch.topk(err.flatten(), 8)
    for v, i in zip(top.values.tolist(), top.indices.tolist()):
        r, t = divmod(i, err.size(1))
        print(f"   err {v:.4f} row {r} (image {r // ns} sample {r % ns}) t {t}  tf {tf[r, t]:.4f} roll {roll[r, t]:.4f} tok {int(rows[r, t])}")
    del m, tr
