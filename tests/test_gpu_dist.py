"""Two-rank execution of the data-parallel trainer on ONE GPU: two processes share cuda:0, rendezvous with `gloo`
on 127.0.0.1, each runs NativeTrainer.xe_step (two steps) and scst_step on its half of the images — with the two-phase backward
and the asynchronous exchange of the decoder half of the gradients (`overlap_allreduce=True`) for the dense class, with ONE
exchange of weights + mask-logit gradients for the supermask class — and the parameters afterwards equal those of the one-rank
full-batch run.  What this does NOT measure is RCCL over xGMI: RCCL refuses two ranks on one device, so the gradient arenas go
through the host here (parallel._StagedWork); the collectives' placement in the step is the product's own.

The rank processes are started as CHILD processes before this process has touched the GPU (conftest.py moves this module to
the front of the session and checks for a GPU with device_count(), which does not initialise it)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _run(args, env):
    return subprocess.Popen([sys.executable, WORKER] + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "supermask"])
def test_two_rank_trainer_on_one_gpu(tmp_path, kind):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29600 + (os.getpid() % 1000) + (7 if kind == "dense" else 13)
    procs = [_run((r, 2, port, tmp_path, kind), env) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    ref = _run((0, 1, port + 1, tmp_path, kind), env)
    out = ref.communicate(timeout=600)[0].decode(errors="replace")
    assert ref.returncode == 0, out
    got, want = np.load(os.path.join(str(tmp_path), f"dp_{kind}.npz")), np.load(os.path.join(str(tmp_path), f"ref_{kind}.npz"))
    # losses: XE steps 1 and 2 and the SCST step (each rank's partial is divided by the GLOBAL normaliser and summed)
    np.testing.assert_allclose(got["losses"], want["losses"], rtol=2e-4, atol=2e-5)
    # parameters after the three updates (fp32 atomics in the weight-gradient GEMMs: summation-order noise only)
    scale = np.abs(want["flat"]).max()
    np.testing.assert_allclose(got["flat"], want["flat"], rtol=0, atol=2e-4 * scale)
    assert want["losses"][0] != want["losses"][1], "the first step moved nothing"
    if kind == "supermask":
        np.testing.assert_allclose(got["masks"], want["masks"], rtol=0, atol=2e-3 * max(1.0, np.abs(want["masks"][np.abs(want["masks"]) < 1e3]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["dense", "supermask"])
def test_rccl_one_rank_group(tmp_path, kind):
    """RCCL under test on the 1-GPU box: `python -m torch.distributed.run --nproc-per-node 1` (the driver's launch line) starts ONE
    rank that forms a real "nccl" (= RCCL) group of one, and the trainer's collectives — the scalar normaliser, the asynchronous
    all-reduce of the decoder half of the gradient arena beside the encoder half of the backward, the encoder half after it (dense), the
    one exchange of weight + mask-logit gradients (supermask) — run through it (parallel.force_collectives).  Losses and parameters
    equal the plain one-process run.  What stays unmeasured is a group of 2-8 GPUs over xGMI: no such node is available."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29700 + (os.getpid() % 1000) + (3 if kind == "dense" else 9)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER, "0", "1", str(port), str(tmp_path), kind, "rccl"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")
    ref = _run((0, 1, port + 1, tmp_path, kind), env)
    out = ref.communicate(timeout=600)[0].decode(errors="replace")
    assert ref.returncode == 0, out
    got, want = np.load(os.path.join(str(tmp_path), f"rccl_{kind}.npz")), np.load(os.path.join(str(tmp_path), f"ref_{kind}.npz"))
    np.testing.assert_allclose(got["losses"], want["losses"], rtol=2e-4, atol=2e-5)
    scale = np.abs(want["flat"]).max()
    np.testing.assert_allclose(got["flat"], want["flat"], rtol=0, atol=2e-4 * scale)


@pytest.mark.gpu
def test_bench_under_torch_distributed_run_one_rank():
    """bench.py launched exactly as the driver launches its ranks (torch.distributed.run, one rank): it forms the RCCL group, runs the
    overlapped XE step (asynchronous split all-reduce on RCCL's stream) and rank 0 prints the one JSON line."""
    import json
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29800 + (os.getpid() % 1000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--overlap-allreduce", "on", "--no-extra-workloads", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["steps"] == 3 and rec["ms_per_step"] > 0 and rec["roofline"]["frac"] > 0
