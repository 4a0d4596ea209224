"""CPU: $DGA_DEFAULT_POLICY is parsed and validated ONCE, in the C library (dga_default_policy), and every front end -- the C entry
points with tiling == NULL, deepgemm_ascend_amd.api, parallel.ExpertShardedGroupedGemm, the deep_gemm_cpp extension -- goes by that
one answer: a typo is refused loudly instead of silently changing the arithmetic (three parsers that disagreed).  The
variable is read once per process, so every case is a fresh interpreter.  No reference counterpart (one kernel, one arithmetic:
/root/reference/deep_gemm_ascend/framework/csrc/python_api.cpp:18-36)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(value, code):
    env = dict(os.environ, PYTHONPATH=str(ROOT))
    env.pop("DGA_DEFAULT_POLICY", None)
    if value is not None:
        env["DGA_DEFAULT_POLICY"] = value
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300, cwd=str(ROOT))


NAME = "import ctypes, deepgemm_ascend_amd as d; from deepgemm_ascend_amd import _lib; b = ctypes.create_string_buffer(32); " \
       "print(_lib.lib().dga_default_policy(b, 32), b.value.decode(), d.api.default_policy())"


@pytest.mark.parametrize("value,name", [(None, "bf16_exact"), ("", "bf16_exact"), ("fast", "fast"), ("strict", "strict"),
                                        ("bf16_exact", "bf16_exact"), ("fast_ue8m0", "fast_ue8m0"), ("bf16_exact_ue8m0", "bf16_exact_ue8m0"),
                                        ("auto", "auto")])
def test_every_front_end_reads_the_one_parsed_name(value, name):
    r = _run(value, NAME)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["0", name, name]


def test_a_typo_is_refused_not_guessed():
    r = _run("fsat", "import ctypes; from deepgemm_ascend_amd import _lib; print(_lib.lib().dga_default_policy(None, 0))")
    assert r.returncode == 0 and r.stdout.strip() == "-9", r.stdout + r.stderr[-1000:]      # DGA_E_RANGE
    r = _run("fsat", "import deepgemm_ascend_amd as d; d.api.default_policy()")
    assert r.returncode != 0 and "names no arithmetic policy" in r.stderr
    # the planned-tiling path of an operator call (no policy, no tiling) raises the same way, before any launch
    r = _run("bf16-exact", "import deepgemm_ascend_amd as d; d.api._planned(0, 64, 128, 256, 1, 0, False, False, None)")
    assert r.returncode != 0 and "names no arithmetic policy" in r.stderr


def test_the_sharded_engine_maps_every_policy_it_can_express():
    code = ("import torch, deepgemm_ascend_amd as d\n"
            "from deepgemm_ascend_amd import parallel\n"
            "mk = lambda p: parallel.ExpertShardedGroupedGemm(0, 1, 8, 64, 256, 512, torch.device('cpu'), None, policy=p, compute=lambda *a: None)\n"
            "print([mk(p).shape.policy for p in (None, 'fast', 'fast_ue8m0', 'bf16_exact', 'bf16_exact_ue8m0', 'strict')])\n"
            "for bad in ('auto', 'fsat'):\n"
            "    try:\n"
            "        mk(bad); print('accepted', bad)\n"
            "    except ValueError as e:\n"
            "        print('refused', bad)\n")
    r = _run(None, code)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "[-1, -2, -3, 7, 23, 3]", lines
    assert lines[1:] == ["refused auto", "refused fsat"]
