"""Feature-sharded (multi-process) path: world_size-2 runs over gloo."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(extra, nproc=2, timeout=300, port=29513):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "sharded_worker.py")] + extra
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    if p.returncode != 0:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "sharded_worker_failure.log"), "w") as fh:
            fh.write(p.stdout + "\n==== stderr ====\n" + p.stderr)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    assert p.stdout.count(": ok") == nproc


def test_sharded_selection_schedule_gloo_cpu():
    """Collective schedules of the sharded selections — BatchTopK (lower-bound MIN, histogram SUMs, tie gather, global
    tie order) and per-row TopK (eight 4-bit radix rounds of all-reduced bucket counts, tie-count gather) — against the
    oracle, two CPU ranks over gloo."""
    _launch(["--sim"])


@pytest.mark.gpu
def test_sharded_training_matches_reference_two_ranks():
    """Two ranks (sharing the GPU, gloo + host staging) each hold half of the features; every step equals the
    unsharded reference golden vectors slice by slice."""
    _launch([])


@pytest.mark.gpu
def test_sharded_schedule_through_rccl_single_rank():
    """The same sharded schedule with backend "nccl" (RCCL) on one rank: every exchange point (int64 MIN, int32 SUM
    on workspace views, all-gather into the tie region, fp64 / fp32 SUMs) goes through RCCL on its own stream, ordered
    against the launch stream; results must still equal the reference goldens."""
    _launch(["--backend", "nccl"], nproc=1, port=29517)
