"""Where does the host-reward SCST step (update on the valid positions) spend its wall time?  Host-side pieces timed one by one."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
dev = torch.device("cuda", 0)
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="bf16")
with torch.no_grad():
    for n_, p in model.named_parameters():
        if n_.endswith("generator.proj.weight"): p.mul_(3.0)
        if n_.endswith("generator.proj.bias"): p[3] += float(sys.argv[1]) if len(sys.argv) > 1 else 5.2
model = model.to(dev).train()
B = 256
batch = bench.synth_batch(B, 36, 2048, 10001, 5, 18, 1000, dev)
tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000)
rw = torch.randn(B * 5)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(8):
    t0 = sync()
    _, _, sq, _ = tr.scst_step(batch, lambda s, g: rw, num_samples=5)
    t1 = sync()
    print("   mean sampled length", float((sq != 0).sum(-1).float().mean()), "positions with a target", int(((sq != 0).sum(-1)).clamp(min=1).sum()))
    tr.scst_step(batch, lambda s, g: rw.cuda(), num_samples=5)
    t2 = sync()
    print(f"iter {it}: host reward (valid positions) {1e3 * (t1 - t0):.2f} ms   device reward (padded) {1e3 * (t2 - t1):.2f} ms")
# pieces of the valid-position table build
seq = torch.randint(0, 5, (1280, 18), device=dev)
mask = (seq != 0).float()
for it in range(3):
    t0 = sync()
    pos = torch.arange(1, 19, device=dev, dtype=mask.dtype)
    cl = (mask * pos).amax(1).clamp_(min=1).to(torch.int64).cpu()
    t1 = sync()
    vr = model._valid_rows(cl, 1280, 18, dev)
    t2 = sync()
    print(f"cap_len read-back {1e3 * (t1 - t0):.2f} ms, _valid_rows {1e3 * (t2 - t1):.2f} ms, Mc {vr[2] if vr else None}")
