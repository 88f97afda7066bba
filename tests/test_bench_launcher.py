"""CPU: `python bench.py --gpus N` must start its own ranks (the way the round-end scaling run calls
it), rendezvous, run the tile exchange and print one JSON line from rank 0 -- exercised here over
gloo with bench.py's exchange self-test standing in for the GPU frames -- and must fail fast with
a one-line reason when the host has fewer GPUs than asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(*args, timeout=240):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *args], cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_starts_its_own_ranks_and_relays_rank_zero(ranks):
    proc = run("--gpus", str(ranks), "--steps", "4", "--selftest-exchange")
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    line = json.loads(lines[0])
    assert line["ok"] is True and line["rccl_ranks"] == ranks and line["frames_reassembled"] == 4


def test_bench_with_more_gpus_than_the_host_has_fails_fast():
    import torch
    have = torch.cuda.device_count()
    proc = run("--gpus", str(have + 2), "--steps", "1", "--warmup", "0", timeout=120)
    assert proc.returncode not in (0, None)
    assert f"needs {have + 2} GPUs" in proc.stderr and proc.stdout.strip() == ""


def test_rank_count_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--selftest-exchange"], cwd=ROOT, env=env, timeout=120,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert proc.returncode != 0 and "started 2 rank(s)" in proc.stderr
