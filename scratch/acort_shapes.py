"""XE step at the ACORT model sizes (commands_acort.sh: d_model 512 / 256 / 104, shared layers, radix vocabulary, T = 26)."""
import sys, time
sys.path[:0]=["/root/repo"]
import torch
import sparse_image_captioning_amd as P
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
sys.path.insert(0, "/root/repo")
import bench
for d, ff, extra in ((512, 2048, {}), (256, 1024, {}), (104, 416, {}), (512, 2048, dict(share_layer_encoder=(0,0,1,1,2,2), share_layer_decoder=(0,0,1,1,2,2), share_att_encoder="kv", share_att_decoder="kv"))):
    cfg = ort_config(d_model=d, dim_feedforward=ff, vocab_size=771, max_seq_length=26, **extra)
    m = P.get_model("relation_transformer")(cfg, precision="bf16").cuda().train()
    tr = NativeTrainer(m, noamopt_warmup=20000)
    batch = bench.synth_batch(128, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, torch.device("cuda"))
    for _ in range(3): tr.xe_step(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr.xe_step(batch)
    torch.cuda.synchronize()
    print(f"d_model {d} ff {ff} {extra and 'acort-shared' or ''}: {(time.perf_counter()-t0)*100:.2f} ms/step, params {sum(p.numel() for p in m.parameters())/1e6:.1f} M", flush=True)

# decode (beam 5, 256 images) at the same widths
for d, ff in ((512, 2048), (256, 1024)):
    cfg = ort_config(d_model=d, dim_feedforward=ff, vocab_size=771, max_seq_length=26)
    m = P.get_model("relation_transformer")(cfg, precision="bf16").cuda().eval()
    batch = bench.synth_batch(256, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, torch.device("cuda"))
    f = lambda: m(att_feats=batch["att_feats"], boxes=batch["boxes"], att_masks=batch["att_masks"], opt={"beam_size": 5}, mode="sample")
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize()
    print(f"decode d_model {d}: {(time.perf_counter()-t0)*200:.2f} ms per 256 images", flush=True)
