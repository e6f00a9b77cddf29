"""bench.py's launch contract, without a GPU: `python bench.py --gpus N` with no launcher starts the N ranks itself (the
driver's N > 1 command may be exactly that), before the launching process imports torch or touches a device, and rank 0's
single JSON line is what comes out."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env=None):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_starts_n_ranks_and_prints_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = run(["--gpus", "3", "--launch-check"], env)
    assert d == {"launch_check": True, "world": 3, "master": "127.0.0.1", "port": d["port"], "torch_imported": False} and d["port"] > 0
    assert run(["--launch-check"], env)["world"] == 1


def test_under_a_launcher_nothing_is_started_twice():
    """with WORLD_SIZE in the environment (torch.distributed.run) the process is a rank, not a launcher"""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    d = run(["--gpus", "4", "--launch-check"], env)
    assert d["world"] == 4 and d["port"] == 29999


def test_workload_defaults():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--workload", "c4"])
    assert (a.kmers, a.variants, a.b, a.r, a.strong) == (3e9, 8e7, 16, 43, True)
    a = bench.parse_args(["--workload", "c5"])
    assert (a.kmers, a.b, a.r, a.strong) == (1e8, 8, 63, False)
    a = bench.parse_args([])
    assert (a.kmers, a.variants, a.b, a.r, a.gpus) == (1e8, 1e6, 4, 43, 1)
    assert bench.blocks_bytes(10, 20, 30) == 64 * 10 + 36 * 20 + 8 * 30


def test_self_launch_ends_all_ranks_when_one_fails():
    """a rank that exits non-zero takes the others down with it (they would sit in a collective until the backend's timeout)
    and the launcher returns a non-zero code at once, not after the survivors' minute"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-fail-rank", "1"], capture_output=True, text=True, timeout=50, env=env)
    assert r.returncode == 3 and time.time() - t0 < 30
