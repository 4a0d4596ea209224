"""CPU: the host half of libdga_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer.

`make asan` compiles the same sources with `-fsanitize=address,undefined` on the host side (the device code is compiled as
usual: GPU sanitizers are not available on this pool) into build/asan, and tests/sanitize/host_driver.cpp drives every entry
point that is host arithmetic in bulk: operator hooks, tiling / kernel selection on both platform descriptions over edge and
random shapes, the CSV cache on well-formed / reference-format / malformed files, the predictor on its own file and on
truncated copies, the 28-int Config derivation, the sharded forward's layout and plan, the argument checks of the launch entry
points, and the selector from four threads while a fifth reloads the predictor and reopens the cache.  A sanitizer report
aborts the driver.  (ThreadSanitizer was run once by hand on the same driver -- clean -- and is not part of the suite:
it needs its own 20 s build.)"""
import os
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    csrc = ROOT / "deepgemm_ascend_amd" / "csrc"
    r = subprocess.run(["make", "-C", str(csrc), "asan"], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    out = ROOT / "build" / "asan"
    exe = out / "host_driver"
    r = subprocess.run([CLANG, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-fno-omit-frame-pointer", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "sanitize" / "host_driver.cpp"),
                        "-o", str(exe), f"-L{out}", "-ldga_hip", f"-Wl,-rpath,{out}", "-lpthread"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, DGA_SAN_TMP=str(tmp_path), UBSAN_OPTIONS="print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1",
               DGA_SAN_PREDICTOR=str(ROOT / "deepgemm_ascend_amd" / "tuned" / "predictor_mi355x.txt"))
    env.pop("DGA_CACHE_FILE_PATH", None); env.pop("CACHE_FILE_PATH", None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok (0 failed checks)"), (r.stdout + r.stderr)[-4000:]
