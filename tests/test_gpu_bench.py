"""bench.py as the driver runs it: one JSON line with the contract's keys, for step counts that
are not multiples of the graph length, with the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_line_contract():
    out = _run("--gpus", "1", "--steps", "200", "--warmup", "7", "--cpu-seconds", "1")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 200 and out["warmup"] == 7
    assert out["unit"] == "evals/s" and out["higher_is_better"] is True and out["scaling"] == "weak"
    assert out["vs_baseline"] is None and out["data"] == "synthetic"
    assert "workload" in out["config"] and "model" not in out["config"]
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    # 200 evaluations are three replays of the 64-evaluation graph + a graph of 8: kernel rate
    assert out["value"] > 50_000
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 6e6 / r["us_per_eval_events"] / 1e3) < 1e-6 * r["achieved"]
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "evals/s" and c["cores"] >= 1 and c["value"] > 0


def test_bench_tiny_step_counts():
    for steps, warmup in ((5, 1), (1, 0), (64, 64)):
        out = _run("--steps", str(steps), "--warmup", str(warmup), "--no-cpu-baseline", "--no-insitu")
        assert out["steps"] == steps and out["value"] > 0


def _two_ranks(port, script="bench.py"):
    env = dict(os.environ)
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, script),
                           "--gpus", "2", "--steps", "128", "--warmup", "64", "--no-cpu-baseline", "--no-insitu"],
                          capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)


def test_bench_two_ranks_flow():
    """The multi-rank flow of bench.py (fixture broadcast, barriers, max over ranks, rank 0 prints)
    with two ranks forced onto this box's single GPU.  RCCL refuses two ranks on one device, so the
    harness tests/bench_two_ranks_gloo.py swaps bench.py's collectives for gloo (recorded in the line;
    bench.py itself has no such switch)."""
    p = _two_ranks(29517, os.path.join("tests", "bench_two_ranks_gloo.py"))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 128 and out["scaling"] == "weak"
    assert abs(out["value"] - 2 * 1e3 / out["ms_per_step"]) < 1e-6 * out["value"]
    assert "cpu_baseline" not in out  # rank 0 at N = 1 only
    assert out["config"]["collectives"].startswith("gloo (tests/bench_two_ranks_gloo.py")


def test_bench_rccl_failure_is_fatal():
    """bench.py itself, two ranks, one GPU: no RCCL world can form (the second rank has no device),
    and an N > 1 value that is not an RCCL measurement must NOT be printed (non-zero exit)."""
    p = _two_ranks(29519)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")], p.stdout
