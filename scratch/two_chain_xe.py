"""Experiment: the XE step's forward + criterion + backward on the full batch (one stream) against two half batches on two
streams (each with its own workspace, gradient arena and library side stream).  Same math as two data-parallel ranks."""
import ctypes as C
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd import parallel
from sparse_image_captioning_amd.utils.config import ort_config
from sparse_image_captioning_amd.training import NativeTrainer
import bench

L = pkg._lib
L.require_gpu()
dev = torch.device("cuda:0")
config = ort_config(drop_prob_src=0.5, max_seq_length=18)
torch.manual_seed(8888)
model = pkg.get_model("relation_transformer")(config, precision="bf16").to(dev)
model.train()
tr = NativeTrainer(model)
data = bench.synth_batch(256, 36, config.att_feat_size, config.vocab_size, 5, config.max_seq_length, 1, dev)
lib = L.lib()
nparts = int(sys.argv[1]) if len(sys.argv) > 1 else 2


def make(dat):
    tok_w = dat["masks"][:, 1:].contiguous().float()
    batch = tr._batch(dat, tok_w)
    nbytes = lib.ortk_train_workspace_bytes(C.byref(model._ccfg), batch.B, batch.S, batch.R, batch.T)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    g = torch.zeros_like(tr.grads)
    return dict(batch=batch, ws=ws, g=g, tok_w=tok_w, keep=dat)


norm = data["masks"][:, 1:].sum().reshape(1).float()
loss = torch.zeros(1, device=dev)


def fwd_bwd(p, stream, seed):
    sp = C.c_void_p(stream.cuda_stream)
    b, ws, g = p["batch"], p["ws"], p["g"]
    pptr = model._eff_params_ptr(True, seed)
    L.check(lib.ortk_forward(C.byref(model._ccfg), pptr, C.byref(b), L.ptr(ws), ws.numel(), None, 0, 1, seed, sp), "fwd")
    L.check(lib.ortk_loss(C.byref(model._ccfg), C.byref(b), L.ptr(ws), ws.numel(), L.ptr(norm), L.ptr(loss), sp), "loss")
    L.check(lib.ortk_backward(C.byref(model._ccfg), pptr, L.ptr(g), C.byref(b), L.ptr(ws), ws.numel(), 1, seed, sp), "bwd")


def bench_it(parts, streams, n=40):
    def one(seed):
        for p, s in zip(parts, streams):
            with torch.cuda.stream(s):
                fwd_bwd(p, s, seed)
    for i in range(5): one(i + 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): one(i + 10)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


full = [make(data)]
print("full batch, one stream: %.3f ms" % bench_it(full, [torch.cuda.Stream()]))
parts = [make(parallel.shard_batch(data, r, nparts)) for r in range(nparts)]
print("%d parts, one stream each: %.3f ms" % (nparts, bench_it(parts, [torch.cuda.Stream() for _ in range(nparts)])))
one_s = torch.cuda.Stream()
print("%d parts, all on one stream: %.3f ms" % (nparts, bench_it(parts, [one_s] * nparts)))
