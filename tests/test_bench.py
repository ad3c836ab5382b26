"""bench.py's launcher and multi-rank plumbing, without a GPU (`--dry-run`: gloo, no rendering).

What the driver does at round end is `python3 bench.py --gpus N --steps K --warmup W` for N in 1, 2, 4, 8; for N > 1 that
plain command has to start its own ranks as fresh child processes.  The GPU side of the same line is covered by
test_gpu_parity.py::test_bench_line_contract."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, cwd=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, cwd=cwd, timeout=600)


@pytest.mark.parametrize("n", [1, 2, 4, 8])
def test_plain_command_launches_its_own_ranks(n, tmp_path):
    run = _run(["--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run"], cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, run.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1
    cfg = d["config"]
    if n == 1:
        assert d["scaling"] == "weak" and cfg["resolution"] == "1920x1080" and cfg["tiles"] == 120 * 68
    else:
        # BASELINE.json configs[4]: ONE fixed 3840x2160 frame split over the ranks
        assert d["scaling"] == "strong" and cfg["resolution"] == "3840x2160" and cfg["tiles"] == 240 * 135
        assert cfg["parallelism"] == f"tiles/{n}" and cfg["gather_floats_per_rank"] == -(-32400 // n) * 16 * 16 * 3


def test_the_gather_of_the_eight_rank_dry_run_is_sized_by_the_librarys_own_shards(tmp_path):
    """configs[4] at N = 8: the packed-tile buffer every rank contributes to the gather (`gather_floats_per_rank`, padded to the largest
    shard) is what crt_shard_tiles — the tile dealing libcrt.so itself uses for crt_set_shard — hands the ranks for 3840x2160 / tile 16;
    the shards cover the frame exactly once."""
    from caitlynrenderer_amd import tiles
    run = _run(["--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"], cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    d = json.loads([l for l in run.stdout.splitlines() if l.strip()][0])
    shards = [tiles.shard_tiles_of_library(3840, 2160, 16, rank=r, world=8) for r in range(8)]
    counts = [len(x) for x in shards]
    assert sum(counts) == 240 * 135 == d["config"]["tiles"] and max(counts) - min(counts) <= 1
    assert d["config"]["gather_floats_per_rank"] == max(counts) * 16 * 16 * 3
    seen = set()
    for x in shards:
        seen |= {(int(a), int(b)) for a, b in x}
    assert len(seen) == 240 * 135
    for r in range(8):                                   # the python launcher's bookkeeping and the library's deal are the same lists
        assert [tuple(map(int, t)) for t in tiles.local_tiles(3840, 2160, 16, r, 8)] == [tuple(map(int, t)) for t in shards[r]]


def test_weak_scaling_option_keeps_per_rank_pixels(tmp_path):
    run = _run(["--gpus", "4", "--scaling", "weak", "--dry-run"], cwd=str(tmp_path))
    assert run.returncode == 0, run.stderr[-3000:]
    d = json.loads(run.stdout.strip())
    assert d["scaling"] == "weak" and d["config"]["resolution"] == "3840x2160"      # 4 x (1920 x 1080)


def test_a_failing_rank_fails_the_command(tmp_path):
    run = _run(["--gpus", "2", "--dry-run", "--resolution", "12by7"], cwd=str(tmp_path))
    assert run.returncode != 0 and run.stdout.strip() == ""


def test_already_under_a_launcher_it_is_one_of_the_ranks(tmp_path):
    """RANK in the environment (the driver's `python -m torch.distributed.run ... bench.py --gpus N` form): no second launch."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run"], capture_output=True, text=True, cwd=str(tmp_path), timeout=600)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_without_a_gpu_the_real_bench_refuses():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    run = _run(["--steps", "1", "--warmup", "0"])
    assert run.returncode != 0 and "needs a GPU" in run.stderr and run.stdout.strip() == ""
