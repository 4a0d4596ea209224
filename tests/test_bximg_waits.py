"""The hand-placed waits of the bf16-exact image builds, checked in the compiled code.

csrc/gemm_fp8_bf16x_image_kernel.hpp and csrc/gemm_fp8_bf16x_aimage_kernel.hpp issue their fragment reads, image stores and (the
A-image build) their global fetches as inline asm and place `s_waitcnt lgkmcnt / vmcnt` themselves from a compile-time schedule:
the compiler neither sees those loads nor orders their consumers against the waits.  scripts/check_bximg_waits.py compiles the
translation unit to ISA, walks every instantiation's main loop twice and fails if any instruction reads a register whose load
is still counted -- the property the GPU parity tests (tests/test_bf16x_image_gpu.py) can only sample.  No GPU needed: hipcc
cross-compiles gfx950."""
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(not (shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists()), reason="hipcc not installed")
def test_no_instruction_reads_a_register_still_in_flight():
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "check_bximg_waits.py")], capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if "problem(s)" in l]
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert len(lines) == 6 and all(l.endswith(" 0 problem(s)") for l in lines), r.stdout   # 2 image x 2 K-tail + 2 A-image builds
    assert sum("aimage" in l for l in lines) == 2
