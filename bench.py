#!/usr/bin/env python3
"""bench.py — headline benchmark of the ORT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 from a plain shell: this process starts N ranks with
                                                            torch.distributed.run before touching the GPU; under
                                                            torch.distributed.run it is one of the ranks)

Default workload = BASELINE.json configs[1]: ORT dense, bf16 MFMA, 256 images x 5 captions x 36 regions x 2048-d per
GPU, teacher-forcing XE.  One "step" = zero_grad -> forward (dropout on) -> fused criterion -> backward ->
(RCCL all-reduce of the flat gradient arena when N > 1) -> clip + Adam(Noam): the step body of the reference's
scripts/train_transformer.py:65-81.  Inputs are synthetic and resident in HBM before the timed
region.  metric = captions/sec = N * B * 5 / step_time  (the reference's "ex/sec", train_transformer.py:85-91).

Other workloads (same JSON contract):  --workload decode|sparse_decode|sparse_xe|scst  (+ --variant, see WORKLOADS)

The default one-GPU run also times every other BASELINE config in the same process and carries them, compactly, in the line's
`workloads` object (ms per step, captions/s, roofline fraction, dominant kernel, CPU baseline); the prose that used to ride on
the line (parity evidence, kernel descriptions, how the traffic figures are read) lives in profiles/bench_notes.json.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work, SURVEY.md §8(d): dense XE 6.354 GFLOP forward per image (5 captions), x3 for fwd+bwd
GFLOP_FWD_PER_IMAGE = 6.354
PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA (MI355X_MICROARCH.md, Chip-level parameters)
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA
PEAK_HBM_GBS = 8000.0


def synth_batch(B, S, F, V, spi, seq_len, seed, device):
    """Synthetic ObjectRelationCollate batch: ReLU-like features (~30 % zeros), relative boxes,
    all-valid regions, BOS + 8..16 uniform tokens + EOS + PAD."""
    rs = np.random.RandomState(seed)
    att = (rs.gamma(0.5, 2.3, size=(B, S, F)) * (rs.uniform(size=(B, S, F)) < 0.7)).astype(np.float32)
    x0, y0 = rs.uniform(0, 0.7, (B, S)), rs.uniform(0, 0.7, (B, S))
    w, h = rs.uniform(0.03, 0.6, (B, S)), rs.uniform(0.03, 0.6, (B, S))
    boxes = np.stack([x0, y0, np.minimum(x0 + w, 1), np.minimum(y0 + h, 1)], -1).astype(np.float32)
    R = B * spi
    seqs = np.zeros((R, seq_len), np.int64)
    masks = np.zeros((R, seq_len), np.float32)
    lens = rs.randint(8, 17, size=R)
    for r in range(R):
        L = int(min(lens[r], seq_len - 2))
        seqs[r, 0] = 2
        seqs[r, 1:1 + L] = rs.randint(4, V, size=L)
        seqs[r, 1 + L] = 3
        masks[r, :L + 2] = 1
    t = lambda a: torch.from_numpy(a).to(device)
    # att_max_len / cap_len: what the collate function reports from the host-side lists (longest region list; decoder positions
    # of every caption that carry a target = BOS + tokens) — no read-back of the masks per step
    cap_len = torch.from_numpy(np.minimum(lens, seq_len - 2).astype(np.int64) + 1)
    return dict(att_feats=t(att), boxes=t(boxes), att_masks=torch.ones(B, S, device=device), seqs=t(seqs), masks=t(masks),
                att_max_len=S, cap_len=cap_len)


CPU_THREADS = 16        # main() --cpu-threads; profiles/r06_cpu_thread_sweep.txt is the sweep behind the default


def cpu_baseline(workload, cfg_dict, seconds=9.0):
    """The oracle (parity-pinned CPU restatement of the reference path, torch CPU fp32) timed on this host's cores on a bounded
    sample of the same workload.  xe: BASELINE configs[0] EXACTLY — 4 images x 5 captions per step, fwd + bwd + clip + Adam (the
    reference's own CPU-runnable case); decode: beam 5 on 8 images per step; scst: greedy baseline + 5
    multinomial rollouts (eval-mode decode; the reference adds dropout, same cost) + teacher-forced update on 8 images per step.
    `cores` = the threads used, `host_threads` = what the host offers."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import helpers as H
    from oracle import ort_oracle as O
    # torch's intra-op pool degrades past a few dozen threads on the 256-thread GPU host at these operator sizes (4 images per step):
    # profiles/r06_cpu_thread_sweep.txt (scratch/cpu_thread_sweep.py: 4 .. 256 threads: 16 is the fastest, 96 captions/s against 54 at 32 and 0.05 at 256); `cores` reports what was used.
    cores = min(os.cpu_count() or 1, CPU_THREADS)
    torch.set_num_threads(cores)
    cfg = O.OCfg(**{k: v for k, v in cfg_dict.items() if not k.startswith("prune")})
    P = H.torch_state(H.dense_param_shapes(cfg_dict), 8888, requires_grad=True)
    Pd = {k: v.detach() for k, v in P.items()}
    kind = "decode" if workload in ("decode", "sparse_decode") else "scst" if workload == "scst" else "xe"
    B = 4 if kind == "xe" else 8
    b = synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1, "cpu")
    state = {}
    n, t0 = 0, None
    t_start = time.time()

    def update(seqs, weights=None):
        for p in P.values():
            p.grad = None
        logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], seqs, b["att_masks"])
        if weights is None:
            loss = O.xe_loss(logp, seqs[:, 1:], b["masks"][:, 1:])
        else:
            rows = seqs[:, 1:]
            loss = O.reward_loss(logp.gather(2, rows.unsqueeze(2)).squeeze(2), rows, weights)
        loss.backward()
        with torch.no_grad():
            O.adam_clip_step(P, {k: p.grad for k, p in P.items()}, state, 1e-4)

    while True:
        if kind == "xe":
            update(b["seqs"])
            units = B * 5
        elif kind == "decode":
            with torch.no_grad():
                O.beam_search(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"], 5)
            units = B
        else:
            with torch.no_grad():
                O.sample_greedy_or_multinomial(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"])
                seq, _ = O.sample_greedy_or_multinomial(Pd, cfg, b["att_feats"], b["boxes"], b["att_masks"], num_random_sample=5, seed=n)
            rows = seq.reshape(-1, seq.size(-1))
            update(torch.cat([rows.new_full((rows.size(0), 1), 2), rows], 1), torch.linspace(-1, 1, rows.size(0)))
            units = B * 5
        if t0 is None:          # first iteration = warm-up
            t0 = time.time()
            continue
        n += 1
        if time.time() - t0 > seconds or time.time() - t_start > 3 * seconds:
            break
    dt = (time.time() - t0) / max(n, 1)
    what = {"xe": "5 captions each, fwd+bwd+Adam", "decode": "beam-5 decode", "scst": "greedy + 5 rollouts + update"}[kind]
    return {"value": round(units / dt, 2), "unit": "captions/sec", "cores": cores, "host_threads": os.cpu_count() or 1, "kind": "port",
            "images_per_step": B,
            "sample": (f"{n} steps of {B} images ({what}{'; BASELINE configs[0] exactly' if kind == 'xe' else ''}), oracle/ort_oracle.py, "
                       f"torch CPU fp32, {cores} of the host's {os.cpu_count() or 1} threads")}


PMC_TAG = "r06"      # profiles/<tag>_*_pmc_{fetch,write}_size.csv: the PMC passes of the CURRENT kernels
PMC_MANIFEST = f"profiles/{PMC_TAG}_manifest.json"     # {"lib_md5", "git_head", ...}: written by scratch/prof_r06.sh with the passes

# The kernels whose counters `traffic` is read from, by their FULL names (template arguments included) as this build launches them:
# a profile of another instantiation is not a measurement of this one.
STACK_DENSE = "ortk::decoder_stack_kernel<false, 32, false>(ortk::StackArgs)"
STACK_SPARSE = "ortk::decoder_stack_kernel<true, 20, false>(ortk::StackArgs)"
STACK_GATHER = "ortk::decoder_stack_kernel<true, 20, true>(ortk::StackArgs)"
ROLLOUT = "ortk::decoder_stack_tp_kernel<8, true>(ortk::StackArgs)"
WGRAD = "wgrad_group_kernel(WgArgs)"
# (round 6: the epilogue is a template argument — 0..3 lean with dropout / gate instances, 4 / 5 soft-max partials (+ sampling
#  candidates), -1 the general one; the first name is the family's dominant instance and must be in the passes)
GEMM_FWD = tuple(f"gemm_bf16_dma256_kernel<false, false, {e}>(ortk_gemm_args, int, int, int)" for e in (0, 1, 2, 3, -1)) + \
           tuple(f"gemm_bf16_glds_kernel<false, false, false, 4, {e}>(ortk_gemm_args, int, int, int)" for e in (0, 1, 2, 3, 4, 5, -1)) + \
           ("gemm_bf16_dma64_kernel<3>(ortk_gemm_args, int, int, int)", "gemm_bf16_dma64_kernel<8>(ortk_gemm_args, int, int, int)")


def lib_md5():
    import hashlib
    from sparse_image_captioning_amd import _lib
    with open(_lib.LIB_PATH, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def pmc_traffic(kernel, workload, precision, B):
    """HBM bytes per launch of ONE kernel family (`kernel`: "stack" = the decoder stack kernel, "gemm" = the forward-layout
    GEMMs) from the committed PMC passes of THIS command (profiles/README.md): 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction
    for 16-B/lane streaming reads), launch-weighted over the kernel's instances.  Returns (bytes | None, note).  None — with the
    reason in the note, which rides on the line as `traffic_note` — for kernels / workloads / sizes without a committed PMC pass,
    when the passes were not collected from THIS library (profiles/<tag>_manifest.json records the md5 of the libortk.so that ran
    them; it must equal the md5 of the library loaded now), and when the CSVs do not hold the kernel under its full name."""
    import csv
    here = os.path.dirname(os.path.abspath(__file__))
    # (kernel kind, workload[_variant]) -> (kernels by full name, file tag).  "step": every kernel of the step (whole-step bound)
    table = {("stack", "decode"): ((STACK_DENSE,), "decode_stack"), ("stack", "sparse_decode"): ((STACK_SPARSE,), "sparse_decode_stack"),
             ("stack", "sparse_decode_988"): ((STACK_GATHER,), "sparse_decode_988_stack"),
             ("gemm", "xe"): (GEMM_FWD, "xe_b256"), ("wgrad", "xe"): ((WGRAD,), "xe_b256"),
             ("gemm", "sparse_xe"): (GEMM_FWD, "sparse_xe_b256"), ("wgrad", "sparse_xe"): ((WGRAD,), "sparse_xe_b256"),
             ("step", "sparse_xe_kernels"): (None, "sparse_xe_kernels_b256"),
             ("gemm", "scst"): (GEMM_FWD, "scst_b256"), ("wgrad", "scst"): ((WGRAD,), "scst_b256"), ("rollout", "scst"): ((ROLLOUT,), "scst_b256")}
    sized = precision == "bf16" and B == (1024 if "decode" in workload else 256)
    if not sized or (kernel, workload) not in table:
        return None, "no PMC pass committed for this workload / size"
    want, tag = table[(kernel, workload)]
    must = want[0] if want else None
    files = [f"{PMC_TAG}_{tag}_pmc_fetch_size.csv", f"{PMC_TAG}_{tag}_pmc_write_size.csv"]
    try:
        man = json.load(open(os.path.join(here, PMC_MANIFEST)))
    except (OSError, ValueError):
        return None, f"STALE: no {PMC_MANIFEST} (which library the PMC passes ran is unknown)"
    now = lib_md5()
    if man.get("lib_md5") != now:
        return None, f"STALE: PMC passes of libortk.so md5 {str(man.get('lib_md5'))[:8]} (commit {str(man.get('git_head'))[:8]}), loaded md5 {now[:8]}"
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    steps_seen = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    seen = set()
    try:
        for cname, fname in zip(("FETCH_SIZE", "WRITE_SIZE"), files):
            for r in csv.DictReader(open(os.path.join(here, "profiles", fname))):
                if r["counter"] != cname:
                    continue
                if want is None:                      # the whole step: every kernel, per launch of the once-per-step criterion kernel
                    tot[cname] += float(r["total"])
                    if r["kernel"].startswith("xent_reg_kernel"):
                        steps_seen[cname] += int(r["launches"])
                elif r["kernel"] in want:
                    seen.add(r["kernel"])
                    tot[cname] += float(r["total"]); n[cname] += int(r["launches"])
    except (OSError, KeyError, ValueError) as e:
        return None, f"profiles/{files[0]} / {files[1]} unreadable ({type(e).__name__}): no traffic figure"
    if want is None:
        if not steps_seen["FETCH_SIZE"] or not steps_seen["WRITE_SIZE"]:
            return None, f"STALE: profiles/{files[0]} / {files[1]} do not hold a once-per-step kernel to count the steps by"
        kb = 2.0 * tot["FETCH_SIZE"] / steps_seen["FETCH_SIZE"] + tot["WRITE_SIZE"] / steps_seen["WRITE_SIZE"]
        return round(kb * 1024), f"HBM bytes per STEP, all kernels, 2*FETCH_SIZE + WRITE_SIZE from profiles/{files[0]} / {files[1]} (libortk.so md5 {now[:8]})"
    if not n["FETCH_SIZE"] or not n["WRITE_SIZE"] or must not in seen:
        return None, f"STALE: profiles/{files[0]} / {files[1]} do not contain the dominant kernel of this build ({must})"
    kb = 2.0 * tot["FETCH_SIZE"] / n["FETCH_SIZE"] + tot["WRITE_SIZE"] / n["WRITE_SIZE"]
    return round(kb * 1024), (f"HBM bytes per launch, 2*FETCH_SIZE + WRITE_SIZE from profiles/{files[0]} / {files[1]} "
                              f"(separate rocprofv3 --pmc passes of this command on libortk.so md5 {now[:8]}), launch-weighted")


NOTES = "profiles/bench_notes.json"       # parity evidence per dtype, kernel descriptions, how `traffic` is read: prose, not on the line


def launch_command(n, argv, port):
    """The driver's own launch line (one rank per GPU of ONE node, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: required by RCCL on this driver
    return subprocess.call(launch_command(n, argv, port), env=env)


def selftest(rank, world, args):
    """No GPU: checks the launch path end to end (ranks start, rendezvous, collective, ONE JSON line from rank 0)."""
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.tensor([float(rank + 1)])
    if world > 1:
        dist.all_reduce(t)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"selftest": True, "n_gpus": world, "rank_sum": t.item(), "steps": args.steps, "warmup": args.warmup}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# name -> (BASELINE config, one-line description).  `scst` is the reference's estimator (utils/training.py:216-237: eval-mode greedy
# baseline, rollouts drawn and differentiated with every dropout on); `scst_nodrop` the dropout-free variant of rounds 1-3.
WORKLOADS = {
    "xe": "configs[1]: ORT dense, 256 images x 5 captions, teacher-forcing XE fwd+bwd+clip+Adam",
    "xe_fp32": ("configs[1], fp32 parity mode: the mode the 'fp32 XE loss within 1e-4' bar is held in; fp32 storage and accumulation, every "
                "product as six bf16 MFMA partial products of three-way split operands, all three GEMM layouts"),
    "sparse_xe": "configs[2]: ORT 95% supermask-sparse XE step, masked dense GEMMs (the reference's flow)",
    "sparse_xe_kernels": "configs[2]: the same step, forward + data-gradient products as sparse kernels (ortk_spmm), weight gradients dense",
    "sparse_xe_988": "configs[2] at 98.8%: the reference's NNZ 0.7M model, masked dense GEMMs",
    "sparse_xe_988_kernels": "configs[2] at 98.8%: sparse kernels where the measured crossover says they pay (enable_sparse_kernels('auto'))",
    "scst": ("configs[3]: ORT dense SCST, the reference's estimator: eval-mode greedy baseline + 5 multinomial rollouts drawn in TRAIN mode "
             "(dropout on) + teacher-forced update under the same dropout masks"),
    "scst_nodrop": "configs[3] without dropout: eval-mode greedy + 5 rollouts in one decode pass, eval-mode update (NOT the reference's estimator)",
    "scst_hostreward": ("configs[3] as the reference runs it end to end: the reward arrives from the HOST (the CIDEr-D scorer's flow: one "
                        "synchronisation per step) and the sampled captions END (generator scaled x3 with an EOS bias, so that sampled lengths "
                        "look like real captions instead of a random-init model's 18 tokens; mean length on the line): the update runs on the valid positions only"),
    "decode": "ORT dense, cached-KV beam-5 decode, 1024 images (mixed precision)",
    "decode_fp32": "beam-5 decode, 1024 images, fp32 parity mode: ORT dense, cached KV, token-exact vs the reference (fp32 storage and accumulation, products split over the bf16 matrix cores)",
    "sparse_decode": "configs[4]: ORT 95% sparse, cached-KV beam-5 decode, 1024 images, decoder stack kernel on the sparse weight stream",
    "sparse_decode_dense_kernels": "configs[4]: the same decode as dense kernels on zero-filled weights (the reference's flow)",
    "sparse_decode_988": "configs[4] at 98.8%: sparse weight stream, gather form (per-column lists; auto from 98.5% zeros on)",
    "sparse_decode_988_scatter": "configs[4] at 98.8%: sparse weight stream, scatter form (the 95% kernel)",
    "sparse_decode_988_dense_kernels": "configs[4] at 98.8%: dense kernels on zero-filled weights",
}

# per-image forward work of the dense model (6.354 GFLOP in all): encoder + the packed cross-attention K|V projection do
# not depend on the caption positions, the decoder stack + generator scale with the decoder rows actually computed
GFLOP_FWD_ENC, GFLOP_FWD_DEC = 1.680, 4.674
GFLOP_DECODE_PER_IMAGE = 6.62       # beam-5 decode, dense (encoder 1.46 + 5 rows x 18 steps x 54.28 MFLOP + attention)


def run_workload(args, workload, variant, steps, warmup, rank, world, dev, pkg, precision=None):
    """Build the model and the synthetic batch of one workload, time `steps` steps after `warmup` (barrier + synchronize on both
    sides, MAX over ranks) and measure the roofline of its dominant kernel with HIP events in extra, untimed steps.
    `variant`: "" | "kernels" (sparse_xe: sparse products) | "dense_kernels" (sparse_decode: zero-filled dense weights) | "nodrop"
    (scst) | "fp32" (decode) | a "988" prefix (98.8 % instead of 95 % zeros)."""
    from sparse_image_captioning_amd.utils.config import ort_config
    from sparse_image_captioning_amd.training import NativeTrainer
    L = pkg._lib
    precision = precision or ("fp32" if variant == "fp32" else args.precision)
    decode = workload in ("decode", "sparse_decode")
    sparse = workload.startswith("sparse")
    keep = 0.012 if "988" in variant else 0.05
    use_csr = workload == "sparse_xe" and variant.endswith("kernels")
    sstream = workload == "sparse_decode" and not variant.endswith("dense_kernels")
    B = args.batch or (1024 if decode else 256)
    spi, S = 5, args.regions
    # ORT pruning / SCST commands use drop_prob_src 0.1 (resources/commands_pruning.sh:240,265); dense XE default 0.5
    config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=args.max_seq_length)
    torch.manual_seed(8888)     # identical weights (and dropout / mask streams) on every rank
    name = "relation_transformer_prune" if workload == "sparse_xe" else "relation_transformer"
    model = pkg.get_model(name)(config, precision=precision)
    if sparse:
        with torch.no_grad():
            if workload == "sparse_xe":    # mask logits of a converged supermask run: |m| = 6, `keep` positive -> the
                # Bernoulli(sigmoid(m)) samples of the training step keep keep*0.9975 + (1-keep)*0.0025 of the weights (5.2 % / 1.4 %)
                for _, m in model.all_pruning_masks():
                    m.copy_(torch.where(torch.rand_like(m) < keep, torch.full_like(m, 6.0), torch.full_like(m, -6.0)))
            else:                                # decode: dense class on densified pruned weights (eval_model.py:64-88)
                for n_, p in model.named_parameters():
                    if p.dim() >= 2:
                        p.mul_((torch.rand_like(p) < keep).float())
    if workload == "scst" and variant == "hostreward":
        # captions that end: the knobs of golden G1 (tests/golden/common.py: generator x 3 + an EOS bias) on the random-init model
        with torch.no_grad():
            for n_, p in model.named_parameters():
                if n_.endswith("generator.proj.weight"):
                    p.mul_(3.0)
                if n_.endswith("generator.proj.bias"):
                    p[config.eos_token_id] += args.scst_eos_bias
    model = model.to(dev)
    if use_csr:                              # sparse products (ortk_spmm) for the weight blocks where they pay
        model.enable_sparse_kernels("auto" if "988" in variant else 0.9, train=True)
    if sstream:
        # the stack kernel pulls the non-zeros of the decoder weights ("988": auto picks the gather form of the stream from 98.5 % zeros on)
        model.enable_sparse_stream(True if variant == "988_scatter" else "auto" if "988" in variant else True)
    batch = synth_batch(B, S, config.att_feat_size, config.vocab_size, spi, config.max_seq_length, 1000 + rank, dev)

    if decode:
        model.eval()
        opt = {"beam_size": 5}
        if args.decode_streams > 1:
            opt["decode_streams"] = args.decode_streams

        def step():
            model(att_feats=batch["att_feats"], boxes=batch["boxes"], att_masks=batch["att_masks"], opt=opt, mode="sample",
                  att_max_len=batch["att_max_len"])
        units_per_step = B
    elif workload == "scst":
        model.train()
        tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000)
        rw = torch.randn(B * 5, device=dev)
        if variant == "hostreward":
            rw = rw.cpu()          # (scst_step reads the sampled lengths back and runs the update on the valid positions)
        sample_dropout = variant != "nodrop"
        lens = []

        def step():
            _, _, seq_, _ = tr.scst_step(batch, lambda seq, greedy: rw, num_samples=5, baseline="greedy", sample_dropout=sample_dropout)
            if variant == "hostreward" and len(lens) < 4:
                lens.append(float((seq_ != 0).sum(-1).float().mean()))
        units_per_step = B * 5
    else:
        model.train()
        tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000,
                           sparsity_target=(1.0 - keep) if workload == "sparse_xe" else None, max_train_step=100000,
                           overlap_allreduce={"auto": None, "on": True, "off": False}[args.overlap_allreduce],
                           allreduce_dtype=args.allreduce_dtype)
        tr.valid_positions = not args.padded_positions

        def step():
            tr.xe_step(batch)
        units_per_step = B * spi

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = elapsed.item()
    ms_per_step = elapsed / steps * 1e3
    value = world * units_per_step / (elapsed / steps)

    # ---- roofline of the dominant kernel, measured live with HIP events around every launch of extra (untimed) steps on the
    # launch stream: once in the TIMED schedule (ortk_prof_enable(2): side stream on, the figure `achieved` reports) and once
    # with every launch on one stream (enable(1): the kernel in isolation).  EVERY rank runs those steps (they contain the
    # data-parallel collectives); only rank 0 records and reports.
    lib = L.lib()
    key = (4 if precision == "bf16" else 0)
    collected = {}
    for level in (2, 1):
        if rank == 0:
            lib.ortk_prof_enable(level)
        step()
        torch.cuda.synchronize()
        if rank == 0:
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            rec = {}
            for k in (key, key + 1, key + 3, 16, 17, 18):
                lib.ortk_prof_collect(k, C.byref(n), C.byref(ms), C.byref(fl))
                lib.ortk_prof_collect_bytes(k, C.byref(by))
                un = C.c_double()
                lib.ortk_prof_collect_units(k, C.byref(un))
                rec[k] = (n.value, ms.value, fl.value, by.value, un.value)
            collected[level] = rec
            lib.ortk_prof_enable(0)
    if rank != 0:
        return None
    per_key, iso = collected[2], collected[1]
    # fp32 parity mode: its products run as six bf16 MFMA partial products of three-way split operands unless ortk_tuning.f32_split = 0
    f32_split = precision != "bf16" and L.set_tuning()["f32_split"] != 0
    peak = PEAK_BF16_TFLOPS if precision == "bf16" else round(PEAK_BF16_TFLOPS / 6.0, 1) if f32_split else PEAK_F32_TFLOPS
    n0, ms0, fl0, by0, _ = per_key[key]
    ach = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
    ach_iso = iso[key][2] / (iso[key][1] * 1e-3) / 1e12 if iso[key][1] > 0 else 0.0
    wtag = workload + ("_" + variant if variant else "")
    traffic, tnote = pmc_traffic("gemm", wtag, precision, B)
    gemm = {"bound": "mfma", "kernel": "forward-layout GEMMs (gemm_bf16_dma256 / glds / dma64)" if precision == "bf16" else ("forward-layout fp32 GEMMs as split bf16 products (gemm_f32x3_kernel / gemm_f32x3p_kernel)" if f32_split else "gemm_f32_kernel (forward layout)"),
            "achieved": round(ach, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": traffic,
            **({"traffic_note": tnote} if traffic is None and tnote.startswith("STALE") else {}),
            "launches": n0, "avg_us": round(ms0 * 1e3 / max(n0, 1), 1), "isolated_frac": round(ach_iso / peak, 4),
            "alg_gflop_per_launch": round(fl0 / max(n0, 1) / 1e9, 2), "alg_bytes_per_launch": round(by0 / max(n0, 1))}
    stack = None        # decode: the one-launch-per-position decoder stack (key 16), same HIP-event hook
    if per_key[16][0]:
        sn, sms, sfl, sby, _ = per_key[16]
        gbs_k = sby / (sms * 1e-3) / 1e9
        st_traffic, st_note = pmc_traffic("stack" if decode else "rollout", wtag, precision, B)
        stack = {"kernel": ("decoder_stack_kernel<sparse>" if sstream else "decoder_stack_kernel") if decode else "decoder_stack_tp_kernel",
                 "launches": sn, "avg_us": round(sms * 1e3 / sn, 1), "alg_bytes_per_launch": round(sby / sn), "achieved": round(gbs_k, 1),
                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs_k / PEAK_HBM_GBS, 4), "traffic": st_traffic,
                 # (beam search: the cached K / V rows are the UNIQUE rows each pass references — beams share ancestors through the
                 #  ancestry table; the beam step counts them on the device, ortk_prof_collect_bytes adds them in)
                 "kv_rows_counted": "unique" if decode else "per row",
                 **({"traffic_note": st_note} if st_traffic is None and st_note.startswith("STALE") else {})}
    chain = None        # rows-stationary chains of the forward pass (key 17): weights streamed out of L2 per workgroup; HBM-side figure
    if per_key[17][0]:
        cn, cms, cfl, cby, _ = per_key[17]
        chain = {"kernel": "row_chain_kernel / row_chain_wide_kernel", "launches": cn, "avg_us": round(cms * 1e3 / cn, 1), "alg_bytes_per_launch": round(cby / cn),
                 "achieved": round(cby / (cms * 1e-3) / 1e9, 1), "unit": "GB/s", "frac": round(cby / (cms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                 "mfma_tflops": round(cfl / (cms * 1e-3) / 1e12, 1)}
    wgrad = None        # grouped weight gradients (key 18): one launch per layer on the side stream, the largest kernel of a training step by time
    if per_key[18][0]:
        gn, gms, gfl, gby, gun = per_key[18]
        held = min(256.0, gun / gn) / 256.0          # average share of the chip's compute units a launch holds
        g_ach = gfl / (gms * 1e-3) / 1e12
        g_iso = iso[18][2] / (iso[18][1] * 1e-3) / 1e12 if iso[18][1] > 0 else 0.0
        g_traffic, g_note = pmc_traffic("wgrad", wtag, precision, B)
        wgrad = {"kernel": "wgrad_group_kernel (a layer's weight + bias gradients in one launch, side stream, 48-96 workgroups)", "bound": "mfma",
                 "launches": gn, "avg_us": round(gms * 1e3 / gn, 1), "achieved": round(g_ach, 1), "peak": peak, "unit": "TFLOP/s",
                 "frac": round(g_ach / peak, 4), "avg_workgroups": round(gun / gn, 1), "frac_of_held_units": round(g_ach / (peak * held), 4),
                 "isolated_frac": round(g_iso / peak, 4), "alg_gflop_per_launch": round(gfl / gn / 1e9, 2),
                 "alg_bytes_per_launch": round(gby / gn), "traffic": g_traffic,
                 **({"traffic_note": g_note} if g_traffic is None and g_note.startswith("STALE") else {})}
    if decode or use_csr:
        # the sparse step and the cached decode are HBM-bound.  Algorithmic bytes: decode = 25.7 MB per image
        # (self-KV reads 10.5 + cross-KV 8.0 + logits 7.2) + the weights once per step (18 steps: 110.9 MB dense bf16, or 4 bytes
        # per non-zero); sparse XE step = activations of the dense step with 4 bytes per non-zero for the weights.
        nnz_bytes = 55.4e6 * keep * 4
        if decode:
            wbytes = nnz_bytes if sparse else (110.9e6 if precision == "bf16" else 221.8e6)
            algo = B * 25.7e6 + config.max_seq_length * wbytes
        else:
            # forward + data-gradient products of one step: X (M x K) and Y (M x N) once each in bf16, 4 bytes per non-zero;
            # the dense weight-gradient products read their two operands once and add into fp32 (M rows: 9 216 / 21 760)
            Me, Md = B * S, B * spi * (config.max_seq_length - 1)
            prods = ([(Me, 512, 2048, 1), (Me, 1536, 512, 6), (Me, 512, 512, 6), (Me, 2048, 512, 6), (Me, 512, 2048, 6), (Me, 6144, 512, 1),
                      (Md, 1536, 512, 6), (Md, 512, 512, 18), (Md, 2048, 512, 6), (Md, 512, 2048, 6), (Md, 10112, 512, 1)])
            algo = sum(c * (2 * (2 * M * K + 2 * M * N) + (2 * M * K + 2 * M * N + 4 * N * K)) for M, N, K, c in prods) + 3 * nnz_bytes
        gbs = algo / (ms_per_step * 1e-3) / 1e9
        step_traffic, step_note = pmc_traffic("step", wtag, precision, B)
        roofline = {"bound": "hbm", "kernel": "whole step", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": step_traffic, "alg_bytes_per_step": round(algo),
                    **({"traffic_note": step_note} if step_traffic is None and step_note.startswith("STALE") else {})}
        if decode and precision != "bf16":
            # fp32 parity mode: 6.62 GFLOP per image — ten times the HBM time of its bytes: MFMA-bound.  Its projections run
            # on the bf16 matrix cores as six bf16 partial products per fp32 product (ortk_gemm.hip: gemm_f32x3_kernel; ortk_tuning.f32_split),
            # so the peak that bounds it is the bf16 MFMA peak / 6 = 416.7 TF/s of fp32-equivalent work (the fp32 MFMA peak, 157.3 TF/s =
            # 43 ms per 1 024 images, bounded the kernel this replaced: `vs_fp32_mfma_peak`)
            tf = GFLOP_DECODE_PER_IMAGE * B / 1e3 / (ms_per_step * 1e-3)
            split = L.set_tuning()["f32_split"] != 0
            pk = PEAK_BF16_TFLOPS / 6.0 if split else PEAK_F32_TFLOPS
            roofline = {"bound": "mfma", "kernel": "whole step", "achieved": round(tf, 1), "peak": round(pk, 1), "unit": "TFLOP/s",
                        "frac": round(tf / pk, 4), "traffic": None, "alg_gflop_per_step": round(GFLOP_DECODE_PER_IMAGE * B, 1),
                        "hbm_frac": round(gbs / PEAK_HBM_GBS, 4), "vs_fp32_mfma_peak": round(tf / PEAK_F32_TFLOPS, 4),
                        "products": "6 bf16 MFMA partial products per fp32 product (three-way operand split)" if split else "fp32 MFMA"}
        if stack is not None:
            roofline["dominant_kernel"] = stack
    else:
        roofline = gemm
        if stack is not None:
            roofline["rollout_kernel"] = stack
    if chain is not None:
        roofline["chain_kernel"] = chain
    if wgrad is not None:
        roofline["wgrad_kernel"] = wgrad
    if not decode and workload != "scst":
        # work the step EXECUTES: the valid-position decoder skips the padded caption positions (same loss and gradients), so the
        # decoder's share is scaled by the rows it runs; `padded_equivalent` is the reference's (R x 17)-row layout
        vr = batch.get("_valid_rows")
        frac_rows = (vr[2] / float(B * spi * (config.max_seq_length - 1))) if vr else 1.0
        exe = (GFLOP_FWD_ENC + GFLOP_FWD_DEC * frac_rows) * 3 * B / 1e3
        roofline["whole_step"] = {"executed_tflop": round(exe, 3), "tflops": round(exe / (ms_per_step * 1e-3), 1),
                                  "frac": round(exe / (ms_per_step * 1e-3) / peak, 4),
                                  "padded_equivalent_tflop": round(GFLOP_FWD_PER_IMAGE * 3 * B / 1e3, 3)}
    wname = workload + ("_" + variant if variant else "")
    if workload == "scst" and variant == "hostreward" and lens:
        roofline["mean_sampled_length"] = round(sum(lens) / len(lens), 2)
    out = {"metric": "captions/sec", "value": round(value, 1), "unit": "captions/sec", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16" if precision == "bf16" else "f32", "data": "synthetic",
           "config": {"workload": WORKLOADS[wname], "images_per_gpu": B, "regions": S, "captions_per_image": spi,
                      "parallelism": (f"dp{world}" + ("+bf16-allreduce" if args.allreduce_dtype == "bf16" else "")) if world > 1 else "single"},
           "roofline": roofline}
    del model
    torch.cuda.empty_cache()
    return out


def compact(r):
    """One extra workload on the headline's line: its time, throughput and roofline in a few dozen bytes."""
    rf = r["roofline"]
    k = rf.get("dominant_kernel") or rf.get("rollout_kernel")
    # (the one-line description of every workload name is in WORKLOADS here and in profiles/bench_notes.json, not on the line)
    out = {"ms_per_step": r["ms_per_step"], "value": r["value"], "dtype": r["dtype"], "config": r["config"]["workload"].split(":")[0],
           "bound": rf["bound"], "frac": rf["frac"], "achieved": rf["achieved"], "unit": rf["unit"]}
    if k:
        out["kernel"] = {x: k[x] for x in ("kernel", "avg_us", "frac", "traffic", "alg_bytes_per_launch")}
    if "chain_kernel" in rf:
        out["chains"] = {x: rf["chain_kernel"][x] for x in ("launches", "avg_us")}
    if "wgrad_kernel" in rf:
        out["wgrad"] = {x: rf["wgrad_kernel"][x] for x in ("launches", "avg_us", "frac", "frac_of_held_units", "traffic")}
    if rf.get("traffic") is not None and "traffic" not in out:
        out["traffic"] = rf["traffic"]
    if "whole_step" in rf:
        out["whole_step_frac"] = rf["whole_step"]["frac"]
    if "mean_sampled_length" in rf:
        out["mean_sampled_length"] = rf["mean_sampled_length"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100: > 1 s of timed region at 12 ms per step)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="xe", choices=("xe", "sparse_xe", "scst", "decode", "sparse_decode"))
    ap.add_argument("--variant", default="", help="workload variant: kernels | 988 | 988_kernels (sparse_xe), dense_kernels | 988 | "
                    "988_dense_kernels (sparse_decode), nodrop | hostreward (scst), fp32 (decode) — see WORKLOADS")
    ap.add_argument("--allreduce-dtype", default="fp32", choices=("fp32", "bf16"),
                    help="N > 1: gradient arenas cross xGMI in fp32 (default: the sum is exact up to order) or rounded to bf16 (half the bytes)")
    ap.add_argument("--overlap-allreduce", default="auto", choices=("auto", "on", "off"),
                    help="all-reduce the decoder half of the gradients while the encoder half of the backward runs "
                         "(auto: on when more than one rank)")
    ap.add_argument("--csr-kernels", "--sparse-kernels", dest="csr_kernels", action="store_true", help="= --variant kernels (sparse_xe)")
    ap.add_argument("--dense-kernels", action="store_true", help="= --variant dense_kernels (sparse_decode)")
    ap.add_argument("--decode-streams", type=int, default=0,
                    help="decode workloads: decode the batch as this many chunks of images on as many streams (0 = one call)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default 256; decode 1024)")
    ap.add_argument("--max-seq-length", type=int, default=18, help="caption length incl. BOS/EOS (18 = BASELINE; the ACORT commands use 26)")
    ap.add_argument("--regions", type=int, default=36, help="regions per image (36 = BASELINE; real bottom-up features have 10-100)")
    ap.add_argument("--precision", default="bf16", choices=("bf16", "fp32"))
    ap.add_argument("--padded-positions", action="store_true",
                    help="teacher forcing over all 17 positions of every caption as the reference does (default: the decoder runs on "
                         "the valid positions only; same loss and gradients)")
    ap.add_argument("--scst-train-sampling", action="store_true", help="(kept for old command lines: the scst workload now IS train-mode sampling)")
    ap.add_argument("--scst-eos-bias", type=float, default=5.4,
                    help="scst --variant hostreward: EOS logit bias on the x3-scaled random-init generator (5.4: sampled captions of ~13 of 18 "
                         "positions, the length of the XE workload's synthetic captions; the line reports the mean sampled length)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=CPU_THREADS, help="threads of the cpu_baseline leg (default: the fastest of the committed sweep)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run on one GPU: do not add the `workloads` object (BASELINE configs[2], [3], [4] timed in this process)")
    ap.add_argument("--selftest", action="store_true",
                    help="launcher check without a GPU: the ranks form a gloo group, all-reduce their rank ids and rank 0 "
                         "prints one JSON line (tests/test_dist_cpu.py)")
    args = ap.parse_args()
    globals()["CPU_THREADS"] = max(1, args.cpu_threads)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (no HIP
        # call, no torch.cuda.*): the ranks are CHILD processes, this one only relays their exit code.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest:
        return selftest(rank, world, args)
    if world > 1 or "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))       # (a one-rank group too: RCCL under test)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import sparse_image_captioning_amd as pkg
    pkg._lib.require_gpu()

    variant = args.variant or ("kernels" if (args.workload == "sparse_xe" and args.csr_kernels) else (
        "dense_kernels" if (args.workload == "sparse_decode" and args.dense_kernels) else ""))
    if args.workload + ("_" + variant if variant else "") not in WORKLOADS:
        sys.exit(f"bench.py: no workload {args.workload} --variant {variant}")
    out = run_workload(args, args.workload, variant, args.steps, args.warmup, rank, world, dev, pkg)
    if rank == 0:
        out["notes"] = NOTES
        from sparse_image_captioning_amd.utils.config import ORT_DEFAULTS
        base = {}

        def cpu(kind):
            if args.no_cpu_baseline or world != 1:
                return None
            if kind not in base:
                base[kind] = cpu_baseline(kind, dict(ORT_DEFAULTS))
            return base[kind]
        # The other BASELINE configs, timed in this same process with the same contract (fewer steps: the whole default run
        # stays within a couple of minutes).  One GPU only: the driver's scaling runs measure the headline.
        if args.workload == "xe" and not variant and world == 1 and not args.no_extra_workloads and not args.batch:
            extra, cpu_kinds = {}, {}
            for wl, var, st, wu in (("xe", "fp32", 8, 2), ("sparse_xe", "", 20, 3), ("sparse_xe", "kernels", 12, 3), ("sparse_xe", "988", 12, 3),
                                    ("sparse_xe", "988_kernels", 12, 3), ("scst", "", 12, 3), ("scst", "nodrop", 12, 3), ("scst", "hostreward", 12, 3),
                                    ("decode", "", 10, 3), ("decode", "fp32", 3, 1), ("sparse_decode", "", 10, 3),
                                    ("sparse_decode", "dense_kernels", 10, 3), ("sparse_decode", "988", 10, 3),
                                    ("sparse_decode", "988_scatter", 10, 3)):      # (988_dense_kernels: by --variant; it measures what
                                                                                    #  sparse_decode_dense_kernels measures, zeros are zeros)
                c = compact(run_workload(args, wl, var, st, wu, rank, world, dev, pkg))
                kind = "decode" if "decode" in wl else "scst" if wl == "scst" else "xe"
                cb = cpu(kind)
                if cb is not None and kind != "xe":
                    c["cpu_baseline"] = kind                   # -> out["cpu_baselines"][kind]
                    cpu_kinds[kind] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "host_threads": cb["host_threads"],
                                       "images_per_step": cb["images_per_step"], "kind": cb["kind"]}       # ("port": the oracle, bench_notes.json)
                extra[wl + ("_" + var if var else "")] = c
            out["workloads"] = extra
            if cpu_kinds:
                out["cpu_baselines"] = cpu_kinds       # the oracle on the host cores for the decode / scst workloads (the xe figure: `cpu_baseline`)
        out["cpu_baseline"] = cpu("decode" if "decode" in args.workload else "scst" if args.workload == "scst" else "xe")
        print(json.dumps(out, separators=(",", ":")), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
