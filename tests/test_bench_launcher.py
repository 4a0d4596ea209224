"""CPU: `python bench.py --gpus N` starts N ranks itself (the reference's multi-card model is N processes, one per
device: /root/reference/deep_gemm_ascend/benchmark_msprof/main.cpp:24-26, framework/benchmark/benchmark.py:249-253).
The compute is stubbed (--stub: gloo, CPU tensors, a numpy stand-in for the step); what is under test is the launcher,
the rendezvous on 127.0.0.1, the barrier / max-over-ranks timing and the JSON contract."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks():
    r = _run(["--gpus", "2", "--stub", "--steps", "5", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 1
    assert d["config"]["backend"] == "gloo" and d["config"]["parallelism"] == "replicas"
    assert len(d["per_rank_kernel_us"]) == 2 and all(x > 0 for x in d["per_rank_kernel_us"])
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["value"] > 0


def test_single_rank_line_has_the_contract_keys():
    d = _json_line(_run(["--stub", "--steps", "3", "--warmup", "1"]).stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic"


def test_world_size_mismatch_is_an_error_not_a_mislabelled_run():
    r = _run(["--gpus", "2", "--stub", "--steps", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0"}, drop=())
    assert r.returncode == 2
    assert "WORLD_SIZE" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_a_failing_rank_fails_the_launcher():
    r = _run(["--gpus", "2", "--stub", "--steps", "2", "--workload", "no_such_workload"])
    assert r.returncode != 0
