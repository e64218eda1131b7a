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

Other workloads (same JSON contract):  --workload decode|sparse_decode|sparse_xe|scst
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


def cpu_baseline(workload, cfg_dict, seconds=12.0):
    """The oracle (parity-pinned CPU restatement of the reference path, torch CPU fp32) timed on this host's cores
    on a bounded sample of the same workload: 8 images x 5 captions per step (decode: 8 images, beam 5)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import helpers as H
    from oracle import ort_oracle as O
    # torch's intra-op pool degrades badly past ~1 thread per physical core group on the 256-thread GPU host
    # (measured: 256 threads -> 0.2 captions/s); 32 threads is the fastest setting found, and is what `cores` reports.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    cfg = O.OCfg(**{k: v for k, v in cfg_dict.items() if not k.startswith("prune")})
    P = H.torch_state(H.dense_param_shapes(cfg_dict), 8888, requires_grad=True)
    B = 8
    b = synth_batch(B, 36, cfg.att_feat_size, cfg.vocab_size, 5, cfg.max_seq_length, 1, "cpu")
    state = {}
    n, t0 = 0, None
    t_start = time.time()
    while True:
        if workload in ("xe", "sparse_xe", "scst"):
            for p in P.values():
                p.grad = None
            logp = O.forward_logp(P, cfg, b["att_feats"], b["boxes"], b["seqs"], b["att_masks"])
            loss = O.xe_loss(logp, b["seqs"][:, 1:], b["masks"][:, 1:])
            loss.backward()
            with torch.no_grad():
                O.adam_clip_step(P, {k: p.grad for k, p in P.items()}, state, 1e-4)
            units = B * 5
        else:
            with torch.no_grad():
                O.beam_search({k: v.detach() for k, v in P.items()}, cfg, b["att_feats"], b["boxes"], b["att_masks"], 5)
            units = B
        if t0 is None:          # first iteration = warm-up
            t0 = time.time()
            continue
        n += 1
        if time.time() - t0 > seconds or time.time() - t_start > 3 * seconds:
            break
    dt = (time.time() - t0) / max(n, 1)
    return {"value": round(units / dt, 2), "unit": "captions/sec", "cores": cores, "kind": "port",
            "sample": f"{n} steps of {B} images ({'5 captions each, fwd+bwd+Adam' if units != B else 'beam-5 decode'}), "
                      f"oracle/ort_oracle.py on torch CPU fp32, {cores} threads"}


PMC_TAG = "r03"      # profiles/<tag>_*_pmc_{fetch,write}_size.csv: the PMC passes of the CURRENT kernels


def pmc_traffic(kernel, workload, precision, B):
    """HBM bytes per launch of ONE kernel family (`kernel`: "stack" = the decoder stack kernel, "gemm" = the forward-layout
    GEMMs) from the committed PMC passes of THIS command (profiles/README.md): 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction
    for 16-B/lane streaming reads), launch-weighted over the kernel's instances.  Returns (bytes | None, note): None for
    kernels / workloads / sizes without a committed PMC profile, and — loudly — when the committed CSVs do not contain the
    kernels this build launches (a stale profile is not a measurement)."""
    import csv
    here = os.path.dirname(os.path.abspath(__file__))
    if kernel == "stack" and workload in ("decode", "sparse_decode") and precision == "bf16" and B == 1024:
        want = ("decoder_stack_kernel<true" if workload == "sparse_decode" else "decoder_stack_kernel<false",)
        must = want[0]
        tag = "sparse_decode_stack" if workload == "sparse_decode" else "decode_stack"
        files = [f"{PMC_TAG}_{tag}_pmc_fetch_size.csv", f"{PMC_TAG}_{tag}_pmc_write_size.csv"]
    elif kernel == "gemm" and workload == "xe" and precision == "bf16" and B == 256:
        want = ("gemm_bf16_glds_kernel<false, false", "gemm_bf16_dma256_kernel<false, false", "gemm_bf16_dma64_kernel")
        must = "gemm_bf16_dma256_kernel<false, false"
        files = [f"{PMC_TAG}_xe_b256_pmc_fetch_size.csv", f"{PMC_TAG}_xe_b256_pmc_write_size.csv"]
    else:
        return None, "no PMC pass committed for this workload / size"
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}
    seen = set()
    try:
        for cname, fname in zip(("FETCH_SIZE", "WRITE_SIZE"), files):
            for r in csv.DictReader(open(os.path.join(here, "profiles", fname))):
                hit = [w for w in want if w in r["kernel"]]
                if hit and r["counter"] == cname:
                    seen.add(hit[0])
                    tot[cname] += float(r["total"]); n[cname] += int(r["launches"])
    except (OSError, KeyError, ValueError) as e:
        return None, f"profiles/{files[0]} / {files[1]} unreadable ({type(e).__name__}): no traffic figure"
    if not n["FETCH_SIZE"] or not n["WRITE_SIZE"] or must not in seen:
        return None, f"STALE: profiles/{files[0]} / {files[1]} do not contain the dominant kernel of this build ({must})"
    kb = 2.0 * tot["FETCH_SIZE"] / n["FETCH_SIZE"] + tot["WRITE_SIZE"] / n["WRITE_SIZE"]
    return round(kb * 1024), (f"HBM bytes per launch, 2*FETCH_SIZE + WRITE_SIZE from profiles/{files[0]} / {files[1]} "
                              "(separate rocprofv3 --pmc passes of this command), launch-weighted over the same kernels")


# Parity evidence carried by the timed mode (tests/, -m gpu; bars in the test sources)
PARITY = {
    "fp32": "XE loss within 1e-4 of the reference goldens (observed 2e-6), every gradient within 2e-4*scale, greedy / beam-3 / "
            "beam-5 tokens exact (tests/test_gpu_model.py: *_vs_reference_golden, full size: test_full_size_config1_*)",
    "bf16": "timed mode: bf16 MFMA operands, fp32 accumulate / soft-max / LayerNorm / residual / optimizer.  Checked against the "
            "library's own fp32 parity mode (HIP-bf16 vs HIP-fp32, not against the oracle directly; the fp32 mode is what the "
            "oracle pins): XE loss within 2e-2 of the golden, gradients within 5 % relative L2 of the fp32 path "
            "(test_mixed_precision_gradients_track_fp32_gradients), teacher-forced log-prob error of the decode <= 0.1 over 64 "
            "images (test_bf16_decode_logprob_bound; the stack kernels: teacher-forced fp32 log-probs of their own tokens within 0.02, "
            "test_decoder_stack_kernel_vs_fp32_and_unfused_executor, test_column_split_stack_kernel_vs_fp32_and_plain_stack), "
            "valid-position layout == padded layout "
            "(test_valid_position_decoder_equals_padded_layout), train-mode dropout replayed through the oracle "
            "(test_train_mode_dropout_vs_oracle, fp32), bench-size determinism / permutation / fused-criterion properties "
            "(test_xe_step_at_bench_size_properties)",
}


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


WORKLOAD_NAMES = {
    "xe": "ORT dense, batch 256 images x 5 captions, teacher-forcing XE fwd+bwd+Adam (BASELINE configs[1])",
    "sparse_xe": "ORT 95% supermask-sparse, batch 256, teacher-forcing XE (BASELINE configs[2]; masked dense GEMM = the reference's flow)",
    "sparse_xe_kernels": ("ORT 95% supermask-sparse, batch 256, teacher-forcing XE (BASELINE configs[2]; forward and data "
                          "gradients as sparse products, weight gradients dense)"),
    "scst": "ORT dense SCST: greedy + 5 multinomial rollouts + teacher-forced update (BASELINE configs[3])",
    "decode": "ORT dense, cached-KV beam-5 decode, 1024 images",
    "sparse_decode": ("ORT 95% sparse, cached-KV beam-5 decode, 1024 images (BASELINE configs[4]; decoder stack kernel on the "
                      "sparse weight stream)"),
    "sparse_decode_dense_kernels": ("ORT 95% sparse, cached-KV beam-5 decode, 1024 images (BASELINE configs[4]; dense kernels on "
                                    "zero-filled weights = the reference's flow)"),
}


def run_workload(args, workload, variant, steps, warmup, rank, world, dev, pkg):
    """Build the model and the synthetic batch of one workload, time `steps` steps after `warmup` (barrier + synchronize on both
    sides, MAX over ranks) and measure the roofline of its dominant kernel with HIP events in extra, untimed steps.
    `variant`: "" | "kernels" (sparse_xe: sparse products) | "dense_kernels" (sparse_decode: zero-filled dense weights)."""
    from sparse_image_captioning_amd.utils.config import ort_config
    from sparse_image_captioning_amd.training import NativeTrainer
    L = pkg._lib
    decode = workload in ("decode", "sparse_decode")
    sparse = workload.startswith("sparse")
    use_csr = workload == "sparse_xe" and variant == "kernels"
    sstream = workload == "sparse_decode" and variant != "dense_kernels"
    B = args.batch or (1024 if decode else 256)
    spi, S = 5, args.regions
    # ORT pruning / SCST commands use drop_prob_src 0.1 (resources/commands_pruning.sh:240,265); dense XE default 0.5
    config = ort_config(drop_prob_src=0.5, prune_type="supermask", max_seq_length=args.max_seq_length)
    torch.manual_seed(8888)     # identical weights (and dropout / mask streams) on every rank
    name = "relation_transformer_prune" if workload == "sparse_xe" else "relation_transformer"
    model = pkg.get_model(name)(config, precision=args.precision)
    if sparse:
        with torch.no_grad():
            if workload == "sparse_xe":    # mask logits of a converged supermask run: |m| = 6, 5 % positive -> the
                # Bernoulli(sigmoid(m)) samples of the training step keep 0.05*0.9975 + 0.95*0.0025 = 5.2 % of the weights
                for _, m in model.all_pruning_masks():
                    m.copy_(torch.where(torch.rand_like(m) < 0.05, torch.full_like(m, 6.0), torch.full_like(m, -6.0)))
            else:                                # decode: dense class on densified 95 %-pruned weights (eval_model.py:64-88)
                for n_, p in model.named_parameters():
                    if p.dim() >= 2:
                        p.mul_((torch.rand_like(p) < 0.05).float())
    model = model.to(dev)
    if use_csr:                              # sparse products (ortk_spmm_ell) for the >= 90 %-sparse weight blocks
        model.enable_sparse_kernels(0.9, train=True)
    if sstream:
        model.enable_sparse_stream(True)     # the stack kernel pulls the non-zeros of the decoder weights
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

        def step():
            tr.scst_step(batch, lambda seq, greedy: rw, num_samples=5, baseline="greedy", sample_dropout=args.scst_train_sampling)
        units_per_step = B * 5
    else:
        model.train()
        tr = NativeTrainer(model, noamopt_factor=1.0, noamopt_warmup=20000,
                           sparsity_target=0.95 if workload == "sparse_xe" else None, max_train_step=100000,
                           overlap_allreduce={"auto": None, "on": True, "off": False}[args.overlap_allreduce])
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
    key = (4 if args.precision == "bf16" else 0)
    collected = {}
    for level in (2, 1):
        if rank == 0:
            lib.ortk_prof_enable(level)
        step()
        torch.cuda.synchronize()
        if rank == 0:
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            rec = {}
            for k in (key, key + 1, key + 3, 16):
                lib.ortk_prof_collect(k, C.byref(n), C.byref(ms), C.byref(fl))
                lib.ortk_prof_collect_bytes(k, C.byref(by))
                rec[k] = (n.value, ms.value, fl.value, by.value)
            collected[level] = rec
            lib.ortk_prof_enable(0)
    if rank != 0:
        return None
    per_key, iso = collected[2], collected[1]
    stack = None        # decode: the one-launch-per-position decoder stack (key 16), same HIP-event hook
    if decode and per_key[16][0]:
        sn, sms, sfl, sby = per_key[16]
        gbs_k = sby / (sms * 1e-3) / 1e9
        stack = {"kernel": ("decoder_stack_kernel<sparse> (ortk_decstack.hip): all decoder layers of one position, rows stationary, the "
                            "NON-ZEROS of the weights streamed as scatter entries and expanded through LDS" if sstream else
                            "decoder_stack_kernel (ortk_decstack.hip): all decoder layers of one position, rows stationary, weights streamed"),
                 "launches_per_step": sn, "avg_launch_us": round(sms * 1e3 / sn, 1),
                 "algorithmic_bytes_per_launch": round(sby / sn), "achieved": round(gbs_k, 1),
                 "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs_k / PEAK_HBM_GBS, 4),
                 "mfma_tflops": round(sfl / (sms * 1e-3) / 1e12, 1), "ms_per_step": round(sms, 3)}
        # (sparse_decode on the DENSE stream is the decode workload's kernel: its counters are that profile's)
        stack["traffic"], stack["traffic_note"] = pmc_traffic("stack", workload if sstream else "decode", args.precision, B)
    n0, ms0, fl0, by0 = per_key[key]
    peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_F32_TFLOPS
    ach = fl0 / (ms0 * 1e-3) / 1e12 if ms0 > 0 else 0.0
    ach_iso = iso[key][2] / (iso[key][1] * 1e-3) / 1e12 if iso[key][1] > 0 else 0.0
    tot_ms = sum(per_key[k][1] for k in (key, key + 1, key + 3))
    tot_fl = sum(per_key[k][2] for k in (key, key + 1, key + 3))
    traffic, traffic_note = pmc_traffic("gemm", workload, args.precision, B)
    gemm = {"kernel": ("gemm_bf16_dma256_kernel<false,false> / gemm_bf16_glds_kernel<false,false,..> / gemm_bf16_dma64_kernel (k-contiguous operands: forward X*W^T, and dY*W on transposed bf16 weight copies; LDS-DMA pipeline)" if args.precision == "bf16"
                       else "gemm_f32_kernel<false,false> (forward X*W^T)"),
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "schedule": "timed schedule (weight-gradient GEMMs on the side stream beside these launches)",
            "isolated": {"achieved": round(ach_iso, 2), "frac": round(ach_iso / peak, 4),
                         "avg_launch_us": round(iso[key][1] * 1e3 / max(iso[key][0], 1), 2),
                         "schedule": "every launch on one stream"},
            "launches_per_step": n0, "avg_launch_us": round(ms0 * 1e3 / max(n0, 1), 2),
            "algorithmic_gflop_per_launch": round(fl0 / max(n0, 1) / 1e9, 3),
            "all_gemm_layouts": {"tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2) if tot_ms > 0 else 0.0,
                                 "ms_per_step": round(tot_ms, 3)},
            "traffic": traffic, "algorithmic_bytes_per_launch": round(by0 / max(n0, 1)), "traffic_note": traffic_note}
    if decode or use_csr:
        # the 95 %-sparse step and the cached decode are HBM-bound.  Algorithmic bytes: decode =
        # 25.7 MB per image (self-KV reads 10.5 + cross-KV 8.0 + logits 7.2) + the weights once per step (18 steps:
        # 110.9 MB dense bf16, or 4 bytes per non-zero); sparse XE step = activations of the dense step (the bytes
        # ortk_prof_collect_bytes sums over the GEMM launches of one step) with 4 bytes per non-zero for the weights.
        nnz_bytes = 2.77e6 * 4
        if decode:
            algo = B * 25.7e6 + config.max_seq_length * (nnz_bytes if sparse else 110.9e6)
        else:
            # forward + data-gradient products of one step: X (M x K) and Y (M x N) once each in bf16, 4 bytes per non-zero;
            # the dense weight-gradient products read their two operands once and add into fp32 (M rows: 9 216 / 21 760)
            Me, Md = B * S, B * spi * (config.max_seq_length - 1)
            prods = ([(Me, 512, 2048, 1), (Me, 1536, 512, 6), (Me, 512, 512, 6), (Me, 2048, 512, 6), (Me, 512, 2048, 6), (Me, 6144, 512, 1),
                      (Md, 1536, 512, 6), (Md, 512, 512, 18), (Md, 2048, 512, 6), (Md, 512, 2048, 6), (Md, 10112, 512, 1)])
            algo = sum(c * (2 * (2 * M * K + 2 * M * N) + (2 * M * K + 2 * M * N + 4 * N * K)) for M, N, K, c in prods) + 3 * nnz_bytes
        gbs = algo / (ms_per_step * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "whole step (all launches): the path is bandwidth-bound as a whole",
                    "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                    "algorithmic_bytes_per_step": round(algo), "traffic": None,
                    "dense_gemm_launches_of_the_step": gemm}
        if stack is not None:
            roofline["dominant_kernel"] = stack
    else:
        roofline = dict(gemm, bound="mfma")
        roofline = {k: roofline[k] for k in ["bound"] + [k for k in gemm]}
    if not decode:
        # dense-equivalent work of the step (6.354 GFLOP forward per image, x3).  For sparse_xe this is the work
        # the masked DENSE GEMMs execute (and the weight gradients always do); the sparse products touch 5 % of it.
        step_tflop = GFLOP_FWD_PER_IMAGE * 3 * B / 1e3
        if workload != "scst":
            roofline["whole_step"] = {"dense_equivalent_tflop": round(step_tflop, 3),
                                      "achieved_tflops": round(step_tflop / (ms_per_step * 1e-3), 2),
                                      "frac_of_peak": round(step_tflop / (ms_per_step * 1e-3) / peak, 4)}
    wname = workload + ("_" + variant if variant else "")
    out = {"metric": "captions/sec", "value": round(value, 1), "unit": "captions/sec", "n_gpus": world, "steps": steps,
           "warmup": warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
           "config": {"workload": WORKLOAD_NAMES[wname], "images_per_gpu": B, "regions": S, "captions_per_image": spi,
                      "parallelism": f"dp{world}" if world > 1 else "single",
                      "storage": ("fp32 master weights / residual stream / logits; MFMA-operand tensors stored bf16; fp32 accumulate"
                                  if args.precision == "bf16" else "fp32"),
                      "sparse_kernels": (os.environ.get("ORTK_SPARSE_FORMAT", "ell16") + " (ortk_spmm)" if use_csr else
                                         "decoder stack kernel, sparse weight stream (ORTK_DEC_SPARSE_STREAM)" if sstream else None)},
           "roofline": roofline}
    del model
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default 100: > 1 s of timed region at 13 ms per step)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="xe", choices=("xe", "sparse_xe", "scst", "decode", "sparse_decode"))
    ap.add_argument("--overlap-allreduce", default="auto", choices=("auto", "on", "off"),
                    help="all-reduce the decoder half of the gradients while the encoder half of the backward runs "
                         "(auto: on when more than one rank)")
    ap.add_argument("--csr-kernels", "--sparse-kernels", dest="csr_kernels", action="store_true",
                    help="sparse_xe: sparse products (ortk_spmm, sorted-ELL images rebuilt on the device every call) for the forward "
                         "and the data gradients instead of MFMA GEMMs on zero-filled weights (the weight gradients stay dense: the "
                         "straight-through mask gradient needs them at every position)")
    ap.add_argument("--dense-kernels", action="store_true",
                    help="sparse_decode: the reference's flow — dense kernels on zero-filled weights — instead of the sparse weight stream")
    ap.add_argument("--decode-streams", type=int, default=0,
                    help="decode workloads: decode the batch as this many chunks of images on as many streams (0 = one call)")
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default 256; decode 1024)")
    ap.add_argument("--max-seq-length", type=int, default=18, help="caption length incl. BOS/EOS (18 = BASELINE; the ACORT commands use 26)")
    ap.add_argument("--regions", type=int, default=36, help="regions per image (36 = BASELINE; real bottom-up features have 10-100)")
    ap.add_argument("--precision", default="bf16", choices=("bf16", "fp32"))
    ap.add_argument("--padded-positions", action="store_true",
                    help="teacher forcing over all 17 positions of every caption as the reference does (default: the decoder runs on "
                         "the valid positions only; same loss and gradients)")
    ap.add_argument("--scst-train-sampling", action="store_true",
                    help="scst: draw the rollouts in TRAIN mode as the reference does (dropout on, generic kernels, separate greedy pass) "
                         "instead of the default eval-mode rollout + dropout-free update (same policy sampled and differentiated)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="default run on one GPU: do not add the `workloads` object (BASELINE configs[2], [3], [4] timed in this process)")
    ap.add_argument("--selftest", action="store_true",
                    help="launcher check without a GPU: the ranks form a gloo group, all-reduce their rank ids and rank 0 "
                         "prints one JSON line (tests/test_dist_cpu.py)")
    args = ap.parse_args()

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
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if world > 1 else 0)

    import sparse_image_captioning_amd as pkg
    pkg._lib.require_gpu()

    variant = "kernels" if (args.workload == "sparse_xe" and args.csr_kernels) else (
        "dense_kernels" if (args.workload == "sparse_decode" and args.dense_kernels) else "")
    out = run_workload(args, args.workload, variant, args.steps, args.warmup, rank, world, dev, pkg)
    if rank == 0:
        out["parity"] = PARITY
        # The other BASELINE configs, timed in this same process with the same contract (fewer steps: the whole default run
        # stays within a couple of minutes).  One GPU only: the driver's scaling runs measure the headline.
        if args.workload == "xe" and world == 1 and not args.no_extra_workloads and not args.batch:
            extra = {}
            for wl, var, st, wu in (("sparse_xe", "", 30, 5), ("sparse_xe", "kernels", 20, 3), ("scst", "", 20, 3),
                                    ("decode", "", 12, 3), ("sparse_decode", "", 12, 3), ("sparse_decode", "dense_kernels", 12, 3)):
                r = run_workload(args, wl, var, st, wu, rank, world, dev, pkg)
                extra[wl + ("_" + var if var else "")] = {k: r[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "config", "roofline")}
            out["workloads"] = extra
        if not args.no_cpu_baseline and world == 1:
            from sparse_image_captioning_amd.utils.config import ORT_DEFAULTS
            out["cpu_baseline"] = cpu_baseline(args.workload, dict(ORT_DEFAULTS))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
