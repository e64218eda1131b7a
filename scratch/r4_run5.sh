#!/bin/bash
# round 4, GPU call 5: MFMA cross-attention in the decoder stack kernel: parity tests + decode timing A/B (stack_debug 64 = old form)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4; mkdir -p $O
python -m pytest tests/test_gpu_model.py -x -q -k "stack or margins or sparse_weight_stream or sparse_decode or decode_at_bench or bf16_decode or large_batch" 2>&1 | tail -8 > $O/pytest_xmfma.log
tail -4 $O/pytest_xmfma.log
python3 - <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev).eval()
b = Bn.synth_batch(1024, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000, dev)
def t(opt, n=6):
    with torch.no_grad():
        m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=opt, mode="sample")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt=opt, mode="sample")
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for ex in ("stack", "stack_rb20"):
    for dbg in (0, 64, 2):
        print(f"{ex} debug {dbg}: {t({'beam_size': 5, 'executor': ex, 'stack_debug': dbg}):.2f} ms", flush=True)
with torch.no_grad():
    for n_, p in m.named_parameters():
        if p.dim() >= 2: p.mul_((torch.rand_like(p) < 0.05).float())
for dbg in (0, 64, 2):
    print(f"sparse_stream debug {dbg}: {t({'beam_size': 5, 'executor': 'sparse_stream', 'stack_debug': dbg}):.2f} ms", flush=True)
PY
