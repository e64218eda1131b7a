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
