"""bench.py's rank launching, exercised without a GPU.

`python bench.py --gpus N` (the form the driver's single-GPU record shows, BENCH_r02.json.cmd) must never print a line whose
`n_gpus` differs from what was asked for: with no launcher in the environment it starts its own N ranks (child processes of a parent
that has not touched the GPU) and exits with their code; under a launcher whose WORLD_SIZE disagrees it refuses (exit 2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_world_size_mismatch_is_refused():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert r.returncode == 2, (r.returncode, r.stderr[-1000:])
    assert "refusing" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_self_launch_starts_ranks_and_propagates_failure():
    """no GPU here: both child ranks fail at device selection, and the parent must report that (non-zero, no JSON line) instead of
    falling back to one rank.  FLK_BENCH_ECHO_RANKS makes each rank print its RANK / WORLD_SIZE first, proving that N ranks were started."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered on the GPU by tests/test_dp_gpu.py::test_bench_two_ranks_rehearsal")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1",
                        "--frames", "16", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=_env(FLK_BENCH_ECHO_RANKS="1", FLK_DIST_BACKEND="gloo"), cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    seen = sorted(l for l in (r.stdout + r.stderr).splitlines() if l.startswith("[bench] rank "))
    assert seen == ["[bench] rank 0 of 2", "[bench] rank 1 of 2"], r.stdout[-1500:] + r.stderr[-1500:]
