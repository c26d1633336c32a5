"""bench.py --gpus N must start N ranks itself (one process per GPU; the reference's analogue is
`mpirun -np P aligner`, main.cpp:1720-1723, 1041-1055).  CPU: --launch-check lets the ranks report over
gloo without touching a GPU."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return env


def test_gpus_2_starts_two_ranks():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=300, env=_env())
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    # (two processes share the pipe: match by pattern, not by line start)
    reports = re.findall(r"LAUNCH_CHECK rank (\d+) of 2 pid \d+ GPU_MAX_HW_QUEUES=(\w+)", p.stdout)
    assert sorted(r[0] for r in reports) == ["0", "1"], p.stdout
    # the ranks got the hardware-queue setting the overlapped stripes need, before any HIP start-up
    assert all(r[1] == "16" for r in reports), reports
    lines = [json.loads(t) for t in re.findall(r"\{\"launch_check\".*?\}\]\}", p.stdout)]
    assert len(lines) == 1, p.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_ranks"] == 2 and lines[0]["ranks_reported"] == [0, 1]
    # the per-rank diagnostics a measured N > 1 line carries (fill / border-wait / traceback ms, K, schedule): one entry per rank
    sys.path.insert(0, str(ROOT))
    import bench

    assert [r["rank"] for r in lines[0]["per_rank"]] == [0, 1]
    assert all(set(r) == set(bench.PER_RANK_KEYS) for r in lines[0]["per_rank"])
    assert {"fill_ms", "border_wait_ms", "rows_per_lane", "overlapped"} <= set(bench.PER_RANK_KEYS)


def test_rank_count_must_match_gpus():
    """inside torch.distributed.run with a different --gpus: refuse instead of measuring the wrong thing"""
    env = _env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--launch-check"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert p.returncode != 0 and "--gpus 2" in p.stderr


def test_single_rank_launch_check():
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--launch-check"], capture_output=True, text=True, timeout=120, env=_env())
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1
