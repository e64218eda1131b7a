"""A/B of the rows-stationary chains inside the real workloads: XE step, SCST step, 1 024-image beam-5 decode with
ortk_tuning.row_chain = 1 / 0 in one process.  python scratch/chain_ab.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
dev = torch.device("cuda", 0)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)


def timeit(step, n, warm=3):
    for _ in range(warm): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for wl in ("xe",):
    res = {}
    for rc in (0, 1, 2, 3, 0, 1, 2, 3):
        pkg._lib.set_tuning(row_chain=rc)
        torch.manual_seed(8888)
        m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev)
        if wl == "decode":
            m.eval()
            b = Bn.synth_batch(1024, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
            step = lambda: m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt={"beam_size": 5}, mode="sample", att_max_len=36)
            t = timeit(step, 8)
        else:
            m.train()
            b = Bn.synth_batch(256, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
            tr = NativeTrainer(m, noamopt_factor=1.0, noamopt_warmup=20000)
            rw = torch.randn(256 * 5, device=dev)
            step = (lambda: tr.xe_step(b)) if wl == "xe" else (lambda: tr.scst_step(b, lambda s_, g_: rw, num_samples=5))
            t = timeit(step, 20 if wl == "xe" else 10)
        res.setdefault(rc, []).append(t)
        del m
        torch.cuda.empty_cache()
    print(f"{wl}: " + "  ".join(f"row_chain={k}: {min(v):.3f} ms" for k, v in sorted(res.items())), flush=True)
