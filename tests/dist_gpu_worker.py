"""One rank of tests/test_gpu_dist.py: NativeTrainer steps on its half of a batch, two processes sharing cuda:0, `gloo`
rendezvous on 127.0.0.1 (RCCL refuses two ranks on one device; the gradient arenas travel through the host, parallel.py).
Usage: python dist_gpu_worker.py RANK WORLD PORT OUT_DIR MODEL(dense|supermask) [rccl]
`rccl`: started by torch.distributed.run with ONE rank — the group is a real RCCL ("nccl") group of one, the collectives of the data
path run through it (parallel.force_collectives), and the results must equal the plain one-process run."""
import os
import sys

rank, world, port, out_dir, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
rccl = len(sys.argv) > 6 and sys.argv[6] == "rccl"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
if not rccl:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: required by RCCL on this driver

import numpy as np
import torch
import torch.distributed as dist

import common as C
import helpers as H
import sparse_image_captioning_amd as pkg
from sparse_image_captioning_amd import parallel
from sparse_image_captioning_amd.training import NativeTrainer
from sparse_image_captioning_amd.utils.config import Config


def build(kind):
    torch.manual_seed(1234)
    if kind == "dense":
        m = pkg.get_model("relation_transformer")(Config(**C.TINY_CFG))
        m.load_state_dict(H.g1_state(), strict=False)
    else:
        m = pkg.get_model("relation_transformer_prune")(Config(**dict(C.TINY_CFG, prune_type="supermask", prune_supermask_init=0.5)))
        m.load_state_dict(H.g1_state(), strict=False)
    return m.cuda()


def run(kind, data, steps, **kw):
    m = build(kind)
    m.exclusive_gpu = False        # two ranks share this GPU: no spin-waiting decode kernels (ortk.h: ORTK_DEC_SPLIT_SMALL)
    m.eval()                                               # (no dropout: the two-rank sum must equal the full-batch step)
    tr = NativeTrainer(m, noamopt_warmup=10, sparsity_target=0.5 if kind != "dense" else None, max_train_step=10, **kw)
    losses = []
    for _ in range(steps):
        losses.append(float(tr.xe_step(data, train=False)))
    # the reward of a sampled caption depends on its GLOBAL index only (rows are grouped by image: shard = slice)
    off = getattr(run, "row0", 0)
    loss, _, seq, _ = tr.scst_step(data, lambda s_, g_: run.reward[off:off + s_.size(0) * s_.size(1)], num_samples=2, train=False)
    losses.append(float(loss))
    flat = m._flat[:m._n_train].detach().cpu().numpy()
    # attention KEY-projection biases have an analytically zero gradient (soft-max is shift-invariant per query); Adam(eps 1e-9)
    # turns their rounding noise into full-size steps, in the reference too: not comparable, zeroed here
    for e in m._entries:
        if e["name"].endswith("linears.1.bias") and "attn" in e["name"]:
            flat[e["offset"]:e["offset"] + e["numel"]] = 0.0
    masks = m._mask_flat.detach().cpu().numpy() if kind != "dense" else np.zeros(1, np.float32)
    return np.asarray(losses), flat, masks, seq.cpu().numpy()


if __name__ == "__main__":
    torch.cuda.set_device(0)
    full = {k: v.cuda() for k, v in H.torch_batch(C.make_inputs(seed=5, n_img=4, n_reg=12, feat=C.TINY_CFG["att_feat_size"],
                                                               vocab=C.TINY_CFG["vocab_size"], spi=2)).items()}
    run.reward = torch.linspace(-1.0, 1.0, 4 * 2).cuda()
    if rccl:
        # one rank per GPU as the driver launches bench.py: RANK / WORLD_SIZE / MASTER_* come from torch.distributed.run
        assert int(os.environ["WORLD_SIZE"]) == 1 and int(os.environ["RANK"]) == 0
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        assert dist.get_backend() == "nccl"
        parallel.force_collectives = True
        losses, flat, masks, seq = run(kind, full, 2, overlap_allreduce=(kind == "dense"))
        np.savez(os.path.join(out_dir, f"rccl_{kind}.npz"), losses=losses, flat=flat, masks=masks)
        dist.barrier()
        dist.destroy_process_group()
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mine = parallel.shard_batch(full)
        run.row0 = rank * 2 * 2                              # images per rank x samples per image
        losses, flat, masks, seq = run(kind, mine, 2, overlap_allreduce=(kind == "dense"))
        if rank == 0:
            np.savez(os.path.join(out_dir, f"dp_{kind}.npz"), losses=losses, flat=flat, masks=masks)
        dist.barrier()
        dist.destroy_process_group()
    else:
        losses, flat, masks, seq = run(kind, full, 2, overlap_allreduce=False)
        np.savez(os.path.join(out_dir, f"ref_{kind}.npz"), losses=losses, flat=flat, masks=masks)
