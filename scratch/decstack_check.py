"""Decoder-stack kernel vs the unfused executor (both mixed precision): run twice with ORTK_DEC_STACK=0 / 1 and compare.
usage: python scratch/decstack_check.py run <out.pt>   |   python scratch/decstack_check.py cmp a.pt b.pt"""
import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/tests/golden")
import torch


def run(out):
    import importlib
    import common as C
    import helpers as H
    pkg = importlib.import_module("sparse_image_captioning_amd")
    from sparse_image_captioning_amd.utils.config import Config
    model = pkg.get_model("relation_transformer")(Config(**C.FULL_CFG), precision=1)
    model.load_state_dict(H.torch_state(H.dense_param_shapes(C.FULL_CFG), C.G2_SEED), strict=False)
    model = model.cuda().eval()
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in H.torch_batch(C.make_inputs(seed=21, n_img=70, n_reg=36, feat=2048, vocab=10001, spi=1, ragged=True)).items()}
    res = {}
    with torch.no_grad():
        for name, opt in (("beam5", {"beam_size": 5}), ("greedy", {"beam_size": 1}), ("beam3", {"beam_size": 3})):
            seq, lp = model(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=opt, mode="sample")
            res[name] = (seq.cpu(), lp.cpu())
    torch.save(res, out)
    print("saved", out)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    for k in A:
        sa, la = A[k]; sb, lb = B[k]
        same = (sa == sb).all(-1).float().mean().item()
        both = (sa == sb)
        d = (la - lb)[both].abs()
        print(k, "rows identical", round(same, 4), "token agreement", round(both.float().mean().item(), 4), "max |dlogp| on agreeing tokens", d.max().item(), "mean", d.mean().item())


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        cmp(sys.argv[2], sys.argv[3])
