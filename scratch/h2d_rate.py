"""The PCIe-inclusive rates of the two headline workloads (the C-ABI takes DEVICE pointers; the reference's loop hands its collated
host batch to `.cuda()` per step): (a) one batch host -> HBM from pinned memory alone, (b) the step with that copy in line on the
same stream, (c) with the NEXT batch's copy on a second stream beside the step (what a loader with pinned double buffers does).
python scratch/h2d_rate.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd.training import NativeTrainer
from sparse_image_captioning_amd.utils.config import ort_config
dev = torch.device("cuda", 0)
torch.manual_seed(8888)
cfg = ort_config(drop_prob_src=0.5, max_seq_length=18)
KEYS = ("att_feats", "boxes", "att_masks", "seqs", "masks")


def host_copy(b):
    return {k: (v.cpu().pin_memory() if torch.is_tensor(v) and v.is_cuda else v) for k, v in b.items()}


def upload(hb, into, stream=None):
    with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
        for k in KEYS: into[k].copy_(hb[k], non_blocking=True)


def timed(fn, n):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for name, B in (("xe", 256), ("decode", 1024)):
    m = pkg.get_model("relation_transformer")(cfg, precision="bf16").to(dev)
    bufs = [Bn.synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1000 + i, dev) for i in range(2)]
    hb = [host_copy(b) for b in bufs]
    mb = sum(hb[0][k].numel() * hb[0][k].element_size() for k in KEYS) / 1e6
    if name == "xe":
        m.train(); tr = NativeTrainer(m, noamopt_factor=1.0, noamopt_warmup=20000, max_train_step=100000)
        work = lambda b: tr.xe_step(b)
        units = B * 5
    else:
        m.eval()
        def work(b):
            with torch.no_grad(): m(att_feats=b["att_feats"], boxes=b["boxes"], att_masks=b["att_masks"], opt={"beam_size": 5}, mode="sample")
        units = B
    t_copy = timed(lambda: upload(hb[0], bufs[0]), 20)
    t_res = timed(lambda: work(bufs[0]), 20)
    def inline():
        upload(hb[0], bufs[0]); work(bufs[0])
    t_inline = timed(inline, 20)
    side = torch.cuda.Stream()
    state = {"i": 0}
    ev = [torch.cuda.Event(), torch.cuda.Event()]
    def overlapped():
        i = state["i"]; state["i"] ^= 1
        # the other buffer's upload for the NEXT step starts once the step that last read it has been queued behind
        side.wait_stream(torch.cuda.current_stream())
        upload(hb[i ^ 1], bufs[i ^ 1], side); ev[i ^ 1].record(side)
        work(bufs[i])
        torch.cuda.current_stream().wait_event(ev[i ^ 1])
    t_over = timed(overlapped, 20)
    print(f"{name}: batch {mb:.1f} MB pinned -> HBM {t_copy:.2f} ms ({mb / t_copy:.1f} GB/s) | resident {t_res:.2f} ms = {units / t_res * 1e3:.0f}/s | "
          f"copy in line {t_inline:.2f} ms = {units / t_inline * 1e3:.0f}/s | next batch's copy beside the step {t_over:.2f} ms = {units / t_over * 1e3:.0f}/s", flush=True)
