"""bench.py --gpus N starts N ranks by itself when no launcher did: the parent process spawns
`python -m torch.distributed.run` as a child, never initialises the GPU, and forwards the ranks' single JSON line.
CPU rehearsal on gloo with --launch-check (process group + barrier + max-over-ranks timing, no engine work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VR2_DIST_BACKEND"] = "gloo"
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus2_self_launches_two_ranks():
    out = _run(["--gpus", "2", "--steps", "3", "--warmup", "0", "--launch-check", "--model", "tiny"])
    assert out["n_gpus"] == 2 and out["steps"] == 3
    assert out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0       # gloo rehearsal: no RCCL ranks are claimed
    assert len(out["rank_ms_per_step"]) == 2
    assert out["ms_per_step"] == max(out["rank_ms_per_step"])             # the job is as slow as its slowest rank


def test_gpus8_self_launches_eight_ranks():
    """the driver's largest run (N = 8) rehearsed on CPU ranks: rendezvous on 127.0.0.1, eight ranks through the barrier and the
    max-over-ranks timing, one JSON line"""
    out = _run(["--gpus", "8", "--steps", "2", "--warmup", "0", "--launch-check", "--model", "tiny"], {"OMP_NUM_THREADS": "1"})
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak"
    assert len(out["rank_ms_per_step"]) == 8 and out["ms_per_step"] == max(out["rank_ms_per_step"])
    assert out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0


def test_gpus1_does_not_spawn():
    out = _run(["--gpus", "1", "--steps", "2", "--launch-check"])
    assert out["n_gpus"] == 1 and out["dist_backend"] is None and len(out["rank_ms_per_step"]) == 1


def test_explicit_launcher_is_respected():
    """Started under torch.distributed.run (as the driver does for N>1) bench.py must NOT spawn again."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["VR2_DIST_BACKEND"] = "gloo"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--launch-check"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
