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


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no rank environment starts two rank processes itself and relays rank 0's line
    (n_gpus = 2, sharding wired through configure_model).  On a box with fewer GPUs than ranks it refuses unless
    --rehearsal is given (ranks then share the GPU over gloo: a schedule rehearsal, said in the line)."""
    import json
    import torch
    small = ["--features", "2048", "--batch", "512", "--layers", "3", "--d-acts", "128", "--k", "8", "--steps", "3", "--warmup", "2",
             "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    bench = os.path.join(ROOT, "bench.py")
    have2 = torch.cuda.device_count() >= 2
    if not have2:
        p = subprocess.run([sys.executable, bench, "--gpus", "2"] + small, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and "--rehearsal" in p.stderr
    p = subprocess.run([sys.executable, bench, "--gpus", "2"] + ([] if have2 else ["--rehearsal"]) + small, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"].startswith("feature-shard x2")
    assert line["rccl_ranks"] == (2 if have2 else 0)
    assert ("REHEARSAL" in line["config"]["parallelism"]) == (not have2)
    assert line["value"] > 0 and line["config"]["regime"] == "activation lists"


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [
    ("headline batchtopk", []),
    ("config 4: CLT JumpReLU + pre-activation loss", ["--nonlinearity", "jumprelu"]),
    ("config 4 in fp16 with loss scaling", ["--nonlinearity", "jumprelu", "--precision", "16-mixed"]),
    ("config 5: CLT TopK + aux loss", ["--nonlinearity", "topk", "--aux"]),
])
def test_bench_eight_rank_rehearsal_of_the_sharded_regimes(name, extra):
    """`bench.py --gpus 8 --rehearsal` at toy size for the regimes the driver's 8-GPU run can meet (BASELINE.json configs 3, 4
    and 5): eight ranks share this box's GPU(s) and exchange over gloo, so every sharded branch of the step (list exchange,
    JumpReLU losses, aux selection, fp16 flag agreement) runs with world_size 8 before a real node does.  A schedule smoke:
    the line must say REHEARSAL and carry a finite loss."""
    import json
    import math
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("a real 8-GPU node runs the RCCL path itself")
    small = ["--features", "2048", "--batch", "256", "--layers", "3", "--d-acts", "128", "--k", "8", "--steps", "2", "--warmup", "2",
             "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rehearsal"] + small + extra, cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=400)
    assert p.returncode == 0, name + "\n" + p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 8 and "REHEARSAL" in line["config"]["parallelism"] and line["rccl_ranks"] == 0
    assert line["value"] > 0 and math.isfinite(line["config"]["loss"])
