"""Which executor is off at S = 100?  Greedy decode with each, then the fp32 teacher-forced log-prob of the emitted tokens."""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import torch
import common as C
import helpers as H
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import Config

st = H.torch_state(H.dense_param_shapes(C.FULL_CFG), C.G2_SEED)
m16 = pkg.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=1); m16.load_state_dict(st, strict=False); m16 = m16.cuda().eval()
m32 = pkg.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=0); m32.load_state_dict(st, strict=False); m32 = m32.cuda().eval()
for n_reg, ragged in ((36, True), (64, True), (65, False), (100, False), (100, True)):
    b = {k: v.cuda() for k, v in H.torch_batch(C.make_inputs(seed=41, n_img=37, n_reg=n_reg, feat=2048, vocab=10001, spi=1, ragged=ragged)).items()}
    for flag in ("2", "0"):
        os.environ["ORTK_DEC_STACK"] = flag
        with torch.no_grad():
            seq, lp = m16(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt={"beam_size": 1}, mode="sample")
            rows = seq[:, 0]
            tf_in = torch.cat([rows.new_full((rows.size(0), 1), 2), rows], 1)
            logp = m32(att_feats=b["att_feats"], boxes=b["boxes"], seqs=tf_in, att_masks=b["att_masks"])
            ref = logp.gather(2, rows.unsqueeze(2)).squeeze(2)
            valid = rows != 0
            err = (lp[:, 0] - ref)[valid].abs()
        print("S", n_reg, "ragged", ragged, "stack", flag, "max err", round(err.max().item(), 4), "mean", round(err.mean().item(), 5), "regions", int(b["att_masks"].sum(1).min()), int(b["att_masks"].sum(1).max()))
