"""Does the XE step gain from two independent half-batch chains on two streams?  (a throughput probe, not a product path)

    one model + trainer, 256 images per step                     -> ms per step                 (the headline)
    two models + trainers, 128 images each, one stream           -> ms per pair of steps        (what halving the kernels costs)
    the same two, on two streams at the same time                -> ms per pair of steps        (what a micro-batched step could reach)

    python scratch/xe_two_streams.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench   # noqa: E402
import sparse_image_captioning_amd as pkg   # noqa: E402
from sparse_image_captioning_amd.utils.config import ort_config   # noqa: E402
from sparse_image_captioning_amd.training import NativeTrainer   # noqa: E402

dev = torch.device("cuda:0")
config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=18)


def make(B, seed):
    torch.manual_seed(8888)
    model = pkg.get_model("relation_transformer")(config, precision="bf16").to(dev)
    model.train()
    tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000)
    batch = bench.synth_batch(B, 36, config.att_feat_size, config.vocab_size, 5, config.max_seq_length, seed, dev)
    return tr, batch


def timeit(fn, n=30, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


tr, b = make(256, 1000)
print("256 images, one step             : %.2f ms" % timeit(lambda: tr.xe_step(b)), flush=True)
del tr, b
torch.cuda.empty_cache()
t1, b1 = make(128, 1000)
t2, b2 = make(128, 1001)
s2 = torch.cuda.Stream()


def seq():
    t1.xe_step(b1); t2.xe_step(b2)


def par():
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    t1.xe_step(b1)
    with torch.cuda.stream(s2):
        t2.xe_step(b2)
    cur.wait_stream(s2)


print("2 x 128 images, one stream       : %.2f ms" % timeit(seq), flush=True)
print("2 x 128 images, two streams      : %.2f ms" % timeit(par), flush=True)
for ss in (0,):
    pkg._lib.set_tuning(side_stream=ss)
    print("  side_stream=%d: one stream %.2f ms, two streams %.2f ms" % (ss, timeit(seq), timeit(par)), flush=True)
pkg._lib.set_tuning(side_stream=1)
