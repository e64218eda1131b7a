"""XE step (256 images, mixed precision) against ortk_tuning.wgrad_wgs: how many workgroups a small weight-gradient GEMM is split into
along K (384 = tuned with the kernel alone on the chip).  Fewer splits = fewer split-K atomics beside the main queue's kernels.
    python scratch/wgrad_split_ab.py [values...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench   # noqa: E402
import sparse_image_captioning_amd as pkg   # noqa: E402
from sparse_image_captioning_amd.utils.config import ort_config   # noqa: E402
from sparse_image_captioning_amd.training import NativeTrainer   # noqa: E402

dev = torch.device("cuda:0")
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="bf16").to(dev)
model.train()
tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000)
b = bench.synth_batch(256, 36, config.att_feat_size, config.vocab_size, 5, config.max_seq_length, 1000, dev)


def timeit(n=40, w=8):
    for _ in range(w): tr.xe_step(b)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.xe_step(b)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


vals = [int(v) for v in sys.argv[1:]] or [384, 256, 192, 128, 96, 384]
for rep in range(2):
    line = []
    for v in vals:
        pkg._lib.set_tuning(wgrad_wgs=v)
        line.append("%d: %.2f" % (v, timeit()))
    print("wgrad_wgs -> ms per XE step   " + "   ".join(line), flush=True)
pkg._lib.set_tuning(wgrad_wgs=384)
