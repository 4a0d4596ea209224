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


# ---- the line the driver parses: ONE stdout line, compact (round 4's 21 KB line went unparsed: BENCH_r04.json parsed = null)

def test_stdout_is_one_compact_line_and_the_detail_file_holds_the_rest(tmp_path):
    detail = tmp_path / "detail.json"
    r = _run(["--stub", "--steps", "3", "--warmup", "1", "--detail-out", str(detail)])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout            # nothing else on stdout: the driver may read the first or the last line
    assert len(lines[0]) < 4096
    d = json.loads(lines[0])
    assert d["detail"] == str(detail) and d["config"]["policy"] == "bf16_exact"
    full = json.loads(detail.read_text())
    assert full["value"] == d["value"] and full["steps"] == 3


def _full_record():
    """A record as large as a real default run's: round 4's committed 21 KB line re-labelled to this round's keys."""
    sys.path.insert(0, str(ROOT))
    d = json.loads((ROOT / "profiles" / "r04_bench.json").read_text())
    d["config"]["policy"] = "bf16_exact"
    d["roofline"].update({"policy": "bf16_exact", "instruction_peak": 2500.0, "frac_of_instruction_peak": 0.47,
                          "traffic_source": "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE " + "x" * 300,
                          "kernel": "gemm_fp8_blockscaled_nt_kernel<..., MATH = 1> (bf16-exact build; persistent form where every CU gets the same tile count)"})
    d["grouped"]["policy"] = "bf16_exact"
    d["grouped"]["fast"] = {"policy": "fast", "ms_gemm": 0.7, "tok_per_s_gemm_only": 4.6e7, "frac_of_8TBps": 0.74, "note": "y" * 200}
    for row in d["shape_list"]["shapes"]:
        row.setdefault("in_contract_frac", 0.2); row.setdefault("in_contract_parity_ok", True)
    return d


def test_compact_of_a_full_size_record_is_under_4_kb_and_keeps_the_contract():
    sys.path.insert(0, str(ROOT))
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 15000          # the premise: the full record is what went unparsed
    c = bench.compact(dict(full, detail="gpurun_out/bench_detail.json"))
    line = json.dumps(c)
    assert len(line) < bench.COMPACT_LIMIT == 4096, len(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in c, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in c["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c["cpu_baseline"], key
    assert c["config"]["workload"].startswith("dense_4096") and "model" not in c["config"]
    assert c["in_contract"]["policy"] == "bf16_exact" and c["grouped"]["roofline"]["bound"] == "hbm"
    assert c["value"] == full["value"] and c["roofline"]["frac"] == full["roofline"]["frac"]   # numbers copied, never recomputed
