"""CPU suite: `python bench.py --gpus N` as the driver invokes it (no launcher around it) must start its N ranks itself, relay rank 0's
JSON line and return the children's code.  Driven here with BENCH_REHEARSAL=stub: gloo children, the sweep replaced by a stub (the
launch / rank / reduction logic of bench.py is what runs; nothing of the path is computed and the line says so)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=300):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_ranks(n):
    r = _run({"BENCH_REHEARSAL": "stub"}, "--gpus", str(n), "--steps", "5", "--warmup", "1", "--no-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 5 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["latents_total"] == 4096 * n
    assert out["nll_total"] == 4096.0 * n                     # every rank's stub total, summed by the pass's all-reduce
    assert "stub" in out["rehearsal"]


def test_bench_reports_a_failing_rank():
    # a config the stub ranks cannot run (it needs the GPU objects): the children fail, the parent must not exit 0 or print a line
    r = _run({"BENCH_REHEARSAL": "stub"}, "--gpus", "2", "--steps", "2", "--warmup", "0", "--config", "c3grad", timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_bench_refuses_a_world_size_mismatch():
    r = _run({"BENCH_REHEARSAL": "stub", "WORLD_SIZE": "1", "RANK": "0"}, "--gpus", "2", "--steps", "1")
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
