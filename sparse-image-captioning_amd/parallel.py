"""Data-parallel helpers (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI on MI355X,
"gloo" in the CPU tests).  The ORT path shards by IMAGE: a rank keeps B/N images with all their caption rows.
Only two exchanges exist on the data path:
  * the scalar normaliser (sum of the 0/1 token mask) so that every rank divides by the GLOBAL count, as
    LanguageModelCriterion / RewardCriterion do on the full batch (utils/losses.py:15-43);
  * ONE all-reduce (SUM) of the flat gradient arena (222 MB fp32 dense; + mask-logit gradients for supermasks).
"""
import torch
import torch.distributed as dist

# True: run the collectives of the data path even in a ONE-rank group (identity results).  The 1-GPU test box cannot form a larger
# RCCL group (RCCL refuses two ranks on one device), so tests/test_gpu_dist.py::test_rccl_one_rank_group sets this to put
# init_process_group("nccl"), the scalar / arena all-reduces and the asynchronous split all-reduce under test on real RCCL.
force_collectives = False


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force_collectives)


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def sample_row_offset(images_per_rank, rows_per_image, r=None):
    """First GLOBAL decode row of rank r's shard (``ortk_decode_opts.sample_row_offset``): the multinomial draws of an SCST rollout
    are keyed by (seed, position, global row), so N ranks on their image shards sample exactly what one process samples on the
    whole batch (rows_per_image = num_samples, + 1 when the greedy baseline rides in the same decode)."""
    r = rank() if r is None else r
    return int(r) * int(images_per_rank) * int(rows_per_image)


def shard_batch(data, r=None, n=None):
    """Rank r's slice of a collated batch: images [r*B/n, (r+1)*B/n) and their caption rows (rows of `seqs` /
    `masks` are grouped by image: row = image*seq_per_img + k, data/collate.py:133-150)."""
    r = rank() if r is None else r
    n = world() if n is None else n
    B = data["att_feats"].size(0)
    assert B % n == 0, f"batch of {B} images does not split over {n} ranks"
    per = B // n
    out = {}
    spi = data["seqs"].size(0) // B if "seqs" in data and data["seqs"] is not None else 0
    for k, v in data.items():
        if k.startswith("_"):
            continue                 # per-batch caches (the valid-position tables): rebuilt for the shard
        if not torch.is_tensor(v):
            out[k] = v[r * per:(r + 1) * per] if isinstance(v, (list, tuple)) and len(v) == B else v
        elif k in ("seqs", "masks", "cap_len"):          # one row per caption
            out[k] = v[r * per * spi:(r + 1) * per * spi]
        elif v.size(0) == B:
            out[k] = v[r * per:(r + 1) * per]
        else:
            out[k] = v
    return out


def _staged(t):
    """gloo (the CPU backend; also what two ranks that SHARE one GPU have to use: RCCL refuses two ranks on one device) cannot
    be relied on for device tensors on this build: exchange such a tensor through the host."""
    return t.is_cuda and dist.get_backend() == "gloo"


class _Bf16Work:
    """SUM all-reduce of an fp32 arena THROUGH bf16: half the bytes on the links (xGMI rings are per-link bound: 153 GB/s x 7), fp32
    again on arrival.  An option of NativeTrainer (`allreduce_dtype="bf16"`), not the default: every rank's addend is rounded to 8
    bits of mantissa before the sum, which is standard gradient compression but not the reference's arithmetic."""

    def __init__(self, t, async_op):
        self.t = t
        self.h = t.detach().to(torch.bfloat16)
        if _staged(t):
            self.h = self.h.cpu()
        self.work = dist.all_reduce(self.h, op=dist.ReduceOp.SUM, async_op=async_op)

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self.t.copy_(self.h.to(self.t.device, torch.float32))


class _StagedWork:
    """all-reduce of a device tensor through a host copy: the copy out waits for the work queued on the current stream (what
    the RCCL collective does on its own stream), ``wait()`` copies the sum back."""

    def __init__(self, t):
        self.t = t
        self.h = t.detach().cpu()
        self.work = dist.all_reduce(self.h, op=dist.ReduceOp.SUM, async_op=True)

    def wait(self):
        self.work.wait()
        self.t.copy_(self.h)


def reduce_scalar_sum(t):
    """In-place SUM over ranks of a (1,) tensor (the loss normaliser)."""
    if _active():
        if _staged(t):
            _StagedWork(t).wait()
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_async(t, dtype=None):
    """Start an in-place SUM all-reduce of `t` (a contiguous slice of the gradient arena) and return its handle
    (``.wait()`` makes the current stream wait for it), or None when there is nothing to exchange.  With the nccl (= RCCL)
    backend the collective first waits for the work already queued on the current stream, then runs on its own stream."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if dtype == "bf16":
        return _Bf16Work(t, True)
    if _staged(t):
        return _StagedWork(t)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_arena(*arenas, dtype=None):
    """In-place SUM of flat gradient arenas; one collective per arena (each is one contiguous bucket).  dtype="bf16": through bf16."""
    if _active():
        for a in arenas:
            if a is not None:
                if dtype == "bf16" and a.numel() > 1:
                    _Bf16Work(a, False).wait()
                elif _staged(a):
                    _StagedWork(a).wait()
                else:
                    dist.all_reduce(a, op=dist.ReduceOp.SUM)
    return arenas
