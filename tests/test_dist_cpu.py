"""world_size-2 `gloo` test (CPU) of the data-parallel scheme of NativeTrainer: shard by image,
all-reduce the scalar normaliser, all-reduce (SUM) the flat gradient arena.  The per-rank gradient comes from the
oracle (no GPU here); the collectives and the sharding code are the product's own (`parallel.py`)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir, n_img=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    torch.set_num_threads(2 if world <= 2 else 1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import common as C
    import helpers as H
    from oracle import ort_oracle as O
    from sparse_image_captioning_amd import parallel
    cfg = O.OCfg(**{k: v for k, v in C.TINY_CFG.items() if not k.startswith("prune")})
    P = H.g1_state(requires_grad=True)
    full = H.torch_batch(C.make_inputs(seed=5, n_img=n_img, n_reg=12, feat=C.TINY_CFG["att_feat_size"], vocab=C.TINY_CFG["vocab_size"], spi=2))
    mine = parallel.shard_batch(full)
    per = n_img // world
    assert mine["att_feats"].shape[0] == per and mine["seqs"].shape[0] == 2 * per
    assert torch.equal(mine["seqs"], full["seqs"][rank * 2 * per:(rank + 1) * 2 * per])
    assert torch.equal(mine["att_feats"], full["att_feats"][rank * per:(rank + 1) * per])
    # the rollout rows of this shard inside the global decode (NativeTrainer.scst_step: opt["sample_row_offset"])
    assert parallel.sample_row_offset(per, 5) == rank * per * 5 and parallel.sample_row_offset(per, 5 + 1) == rank * per * 6
    norm = mine["masks"][:, 1:].sum().reshape(1).clone()
    parallel.reduce_scalar_sum(norm)
    assert abs(norm.item() - full["masks"][:, 1:].sum().item()) < 1e-6
    logp = O.forward_logp(P, cfg, mine["att_feats"], mine["boxes"], mine["seqs"], mine["att_masks"])
    loss = -(logp.gather(2, mine["seqs"][:, 1:].unsqueeze(2)).squeeze(2) * mine["masks"][:, 1:]).sum() / norm[0]
    loss.backward()
    names = sorted(P)
    arena = torch.cat([P[n].grad.reshape(-1) for n in names])
    loss_t = loss.detach().reshape(1).clone()
    # the trainer's overlapped exchange: the tail ("decoder half") starts asynchronously, the head follows, then wait
    split = arena.numel() // 3
    pending = parallel.allreduce_async(arena[split:])
    parallel.allreduce_arena(arena[:split], loss_t)
    pending.wait()
    # the bf16 exchange option (NativeTrainer(allreduce_dtype="bf16")): same sum to 8 bits of mantissa per addend
    a16 = torch.cat([P[n].grad.reshape(-1) for n in names])
    p16 = parallel.allreduce_async(a16[split:], dtype="bf16")
    parallel.allreduce_arena(a16[:split], dtype="bf16")
    p16.wait()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), arena=arena.numpy(), loss=loss_t.numpy(), arena_bf16=a16.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_equals_full_batch(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    import common as C
    import helpers as H
    from oracle import ort_oracle as O
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    cfg = O.OCfg(**{k: v for k, v in C.TINY_CFG.items() if not k.startswith("prune")})
    P = H.g1_state(requires_grad=True)
    full = H.torch_batch(C.make_inputs(seed=5, n_img=4, n_reg=12, feat=C.TINY_CFG["att_feat_size"], vocab=C.TINY_CFG["vocab_size"], spi=2))
    logp = O.forward_logp(P, cfg, full["att_feats"], full["boxes"], full["seqs"], full["att_masks"])
    loss = O.xe_loss(logp, full["seqs"][:, 1:], full["masks"][:, 1:])
    loss.backward()
    ref = torch.cat([P[n].grad.reshape(-1) for n in sorted(P)]).numpy()
    assert abs(float(got["loss"][0]) - loss.item()) < 1e-5
    np.testing.assert_allclose(got["arena"], ref, rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(got["arena_bf16"], ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())       # bf16 addends


def test_eight_rank_shard_arithmetic_equals_full_batch(tmp_path):
    """The node's real rank count (BASELINE configs[3]: 8 GPUs): 8 gloo ranks x 2 images.  Each rank takes its images and captions,
    the global normaliser is the all-reduced mask sum, `sample_row_offset` places its rollout rows in the global decode, and the summed
    loss / gradient arena (split asynchronous all-reduce, as the trainer's overlapped exchange issues it) equal the full-batch step's."""
    port = 29500 + ((os.getpid() + 1234) % 2000)
    mp.spawn(_worker, args=(8, port, str(tmp_path), 16), nprocs=8, join=True)
    import common as C
    import helpers as H
    from oracle import ort_oracle as O
    got = np.load(os.path.join(str(tmp_path), "dp.npz"))
    cfg = O.OCfg(**{k: v for k, v in C.TINY_CFG.items() if not k.startswith("prune")})
    P = H.g1_state(requires_grad=True)
    full = H.torch_batch(C.make_inputs(seed=5, n_img=16, n_reg=12, feat=C.TINY_CFG["att_feat_size"], vocab=C.TINY_CFG["vocab_size"], spi=2))
    logp = O.forward_logp(P, cfg, full["att_feats"], full["boxes"], full["seqs"], full["att_masks"])
    loss = O.xe_loss(logp, full["seqs"][:, 1:], full["masks"][:, 1:])
    loss.backward()
    ref = torch.cat([P[n].grad.reshape(-1) for n in sorted(P)]).numpy()
    assert abs(float(got["loss"][0]) - loss.item()) < 1e-5
    np.testing.assert_allclose(got["arena"], ref, rtol=1e-4, atol=4e-6)
    np.testing.assert_allclose(got["arena_bf16"], ref, rtol=2e-2, atol=2e-2 * np.abs(ref).max())


def _scst_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import common as C
    import helpers as H
    from oracle import ort_oracle as O
    from sparse_image_captioning_amd import parallel
    cfg = O.OCfg(**{k: v for k, v in C.TINY_CFG.items() if not k.startswith("prune")})
    P = H.g1_state(requires_grad=True)
    Pd = {k: v.detach() for k, v in P.items()}
    full = H.torch_batch(C.make_inputs(seed=5, n_img=4, n_reg=12, feat=C.TINY_CFG["att_feat_size"], vocab=C.TINY_CFG["vocab_size"], spi=1))
    full = {k: v for k, v in full.items() if k not in ("seqs", "masks")}
    mine = parallel.shard_batch(full)
    B, ns, seed = mine["att_feats"].size(0), 3, 77
    assert B == 2
    # NativeTrainer.scst_step: the rollout's draws are keyed by the GLOBAL row (parallel.sample_row_offset -> opt["sample_row_offset"])
    off = parallel.sample_row_offset(B, ns)
    assert off == rank * B * ns
    with torch.no_grad():
        seq, _ = O.sample_greedy_or_multinomial(Pd, cfg, mine["att_feats"], mine["boxes"], mine["att_masks"], num_random_sample=ns,
                                                seed=seed, sample_row_offset=off)
    rows = seq.reshape(-1, seq.size(-1))
    mask = (rows != 0).float()
    reward = torch.linspace(-1.0, 1.0, 4 * ns)[off:off + B * ns]             # the global reward vector's slice of this shard
    norm = mask.sum().reshape(1).clone()
    parallel.reduce_scalar_sum(norm)                                        # RewardCriterion over the GLOBAL batch
    tf_in = torch.cat([rows.new_full((rows.size(0), 1), C.BOS), rows], 1)
    logp = O.forward_logp(P, cfg, mine["att_feats"].repeat_interleave(ns, 0), mine["boxes"].repeat_interleave(ns, 0), tf_in,
                          mine["att_masks"].repeat_interleave(ns, 0))
    tok = logp.gather(2, rows.unsqueeze(2)).squeeze(2)
    loss = -(tok * mask * reward[:, None]).sum() / norm[0]
    loss.backward()
    arena = torch.cat([P[n].grad.reshape(-1) for n in sorted(P)])
    loss_t = loss.detach().reshape(1).clone()
    parallel.allreduce_arena(arena, loss_t)
    gathered = [torch.zeros_like(rows) for _ in range(world)]
    dist.all_gather(gathered, rows)
    if rank == 0:
        np.savez(os.path.join(out_dir, "scst.npz"), arena=arena.numpy(), loss=loss_t.numpy(), rows=torch.cat(gathered).numpy(), norm=norm.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_scst_sharding_samples_and_updates_like_one_process(tmp_path):
    """SCST shards by image too: each rank rolls out ITS images with the multinomial draws keyed by the
    global decode row (`parallel.sample_row_offset`, what `NativeTrainer.scst_step` puts into `opt["sample_row_offset"]`), divides
    by the GLOBAL mask sum and adds its gradient arena into the all-reduce.  Two gloo ranks (the oracle stands in for the GPU path)
    sample token for token what ONE process samples on the whole batch, and the summed loss / gradient arena equal the
    full-batch RewardCriterion's (utils/losses.py:15-29, utils/training.py:239-255)."""
    port = 29500 + ((os.getpid() + 777) % 2000)
    mp.spawn(_scst_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    import common as C
    import helpers as H
    from oracle import ort_oracle as O
    got = np.load(os.path.join(str(tmp_path), "scst.npz"))
    cfg = O.OCfg(**{k: v for k, v in C.TINY_CFG.items() if not k.startswith("prune")})
    P = H.g1_state(requires_grad=True)
    Pd = {k: v.detach() for k, v in P.items()}
    full = H.torch_batch(C.make_inputs(seed=5, n_img=4, n_reg=12, feat=C.TINY_CFG["att_feat_size"], vocab=C.TINY_CFG["vocab_size"], spi=1))
    ns = 3
    with torch.no_grad():
        seq, _ = O.sample_greedy_or_multinomial(Pd, cfg, full["att_feats"], full["boxes"], full["att_masks"], num_random_sample=ns, seed=77)
    rows = seq.reshape(-1, seq.size(-1))
    np.testing.assert_array_equal(got["rows"], rows.numpy())
    assert len({tuple(r) for r in rows.tolist()}) > 4                       # (the rollouts really differ from row to row)
    tf_in = torch.cat([rows.new_full((rows.size(0), 1), C.BOS), rows], 1)
    logp = O.forward_logp(P, cfg, full["att_feats"].repeat_interleave(ns, 0), full["boxes"].repeat_interleave(ns, 0), tf_in,
                          full["att_masks"].repeat_interleave(ns, 0))
    loss = O.reward_loss(logp.gather(2, rows.unsqueeze(2)).squeeze(2), rows, torch.linspace(-1.0, 1.0, 4 * ns))
    loss.backward()
    ref = torch.cat([P[n].grad.reshape(-1) for n in sorted(P)]).numpy()
    assert abs(float(got["norm"][0]) - float((rows != 0).sum())) < 1e-6
    assert abs(float(got["loss"][0]) - loss.item()) < 1e-5
    np.testing.assert_allclose(got["arena"], ref, rtol=1e-4, atol=2e-6)


def test_shard_batch_single_process():
    from sparse_image_captioning_amd import parallel
    data = dict(att_feats=torch.zeros(6, 3, 4), boxes=torch.zeros(6, 3, 4), att_masks=torch.ones(6, 3),
                seqs=torch.arange(6 * 5 * 18).view(30, 18), masks=torch.ones(30, 18), image_ids=list(range(6)),
                cap_len=torch.arange(30), att_max_len=3, _valid_rows=("cached", "tables"))
    s = parallel.shard_batch(data, 2, 3)
    assert s["att_feats"].shape[0] == 2 and s["image_ids"] == [4, 5]
    assert torch.equal(s["seqs"], data["seqs"][20:30])
    assert torch.equal(s["cap_len"], torch.arange(20, 30)) and s["att_max_len"] == 3 and "_valid_rows" not in s
    with pytest.raises(AssertionError):
        parallel.shard_batch(data, 0, 4)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` from a plain shell (the form the driver's scaling run uses when WORLD_SIZE is unset) starts N
    ranks itself — torch.distributed.run children, rendezvous on 127.0.0.1 — and rank 0 prints ONE JSON line with n_gpus = N.
    Checked without a GPU through the gloo self-test of the same launch path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--selftest"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec == {"selftest": True, "n_gpus": 2, "rank_sum": 3.0, "steps": 3, "warmup": 1}
    # a mismatch between --gpus and the group size is refused, not silently run
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--selftest"],
                         env=dict(env, WORLD_SIZE="2", RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "20"], 29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd and cmd[-4:] == ["--gpus", "8", "--steps", "20"]
