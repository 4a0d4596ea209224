"""CPU: compile-time guards on the HIP kernels (hipcc cross-compiles for gfx950 without a GPU).
A register spill in the MFMA pipeline is a silent 1.6x slowdown (seen in r01 when a runtime branch was added
inside the software pipeline), so no kernel may spill or use scratch."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "deepgemm_ascend_amd" / "csrc"


def _ship_flags(unit):
    """The flags the Makefile compiles this unit with, taken from `make -n` (not a hand-kept copy: a unit's own switches --
    FLAGS_<unit>, NOFORM_<unit> -- change what ships)."""
    obj = f"../../build/csrc/{Path(unit).stem}.o"
    r = subprocess.run(["make", "-n", "-B", "-C", str(CSRC), obj], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr[-1000:]
    line = [l for l in r.stdout.splitlines() if "hipcc" in l and f" {unit} " in l + " "]
    assert line, r.stdout[-1000:]
    words = line[-1].split()
    keep, skip = [], 0
    for i, w in enumerate(words[1:], 1):
        if skip:
            skip -= 1
            continue
        if w in ("-c",) or w == unit:
            continue
        if w == "-o":
            skip = 1
            continue
        keep.append(w)
    return keep


@pytest.mark.parametrize("unit,min_kernels,tile_kernel", [
    ("dga_launch.hip", 4, "gemm_fp8_strict_nt_kernel"),          # strict x 2, element-wise, split-K combine
    ("dga_rows.hip", 5, "pad_rows_kernel"),                      # row copies, routing, the odd-K re-layout pass
    ("dga_launch_menu_e.hip", 10, "gemm_fp8_blockscaled_nt_kernel"),   # bf16-exact builds (5 tiles x k-tail / no k-tail)
    ("dga_launch_menu_a.hip", 6, "gemm_fp8_blockscaled_nt_kernel"),
    ("dga_launch_menu_b.hip", 10, "gemm_fp8_blockscaled_nt_kernel"),
    ("dga_launch_menu_c.hip", 28, "gemm_fp8_blockscaled_nt_kernel"),
    ("dga_launch_menu_d.hip", 11, "gemm_fp8_blockscaled_nt_"),   # 10 persistent loader-wave builds + the persistent continuous one
    ("dga_diag.hip", 2, "gemm_fp8_blockscaled_nt_kernel"),
    ("dga_launch_menu_f.hip", 6, "gemm_fp8_bf16x_"),            # bf16-exact image builds (8 / 4 waves, A-image) x k-tail
    ("dga_launch_menu_g.hip", 6, "gemm_fp8_wsk_kernel"),         # workgroup split-K (3 row counts x k-tail)
    ("dga_launch_menu_h.hip", 2, "gemm_fp8_bf16x_persistent_kernel"),   # persistent bf16-exact 128x256 build x k-tail
    ("dga_b16.hip", 26, "gemm_b16_"),                            # 16-bit tile builds + the workgroup split-K (3 builds x bf16 / fp16)
    ("dga_b16_w4.hip", 2, "gemm_b16_w4_kernel"),                 # the four-wave 32x32x16 build (bf16 / fp16): accumulators in AGPRs on purpose
    ("dga_launch_menu_i.hip", 20, "gemm_fp8_blockscaled_nt_kernel"),   # hardware-scale builds (MATH = 2): 10 tile builds x k-tail
    ("dga_launch_menu_j.hip", 6, "gemm_fp8_blockscaled_nt_kernel"),    # four-wave builds with AGPR accumulators on purpose (MATH = 2 / 3)
    ("dga_launch_menu_k.hip", 2, "gemm_fp8_blockscaled_nt_streamk_kernel"),   # one-launch Stream-K (promotion / hardware-scale form)
    ("dga_launch_menu_l.hip", 4, "gemm_fp8_bf16x_grouped_kernel"),      # the masked-grouped layout's bf16-exact kernel (k-tail x nt weights)
    ("dga_launch_menu_m.hip", 2, "gemm_fp8_bf16x_streamk_kernel"),      # one-launch Stream-K of the bf16-exact persistent kernel x k-tail
    ("dga_launch_menu_n.hip", 2, "gemm_fp8_bf16x_dsk_kernel")])         # one-launch split-K of the 64 x 128 tile for decode rows x k-tail
def test_no_kernel_spills_or_scratch(unit, min_kernels, tile_kernel):
    flags = _ship_flags(unit)
    assert "--offload-arch=gfx950" in flags and "-O3" in flags
    cmd = ["/opt/rocm/bin/hipcc", *flags, "--cuda-device-only", "-S", "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage", str(CSRC / unit)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=str(CSRC))   # (the Makefile's include paths are relative)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen = None, 0
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            seen += 1
        m = re.search(r"(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m:
            # the persistent builds park some scalars in VGPR lanes (v_writelane / v_readlane): tile-list state of the
            # loader waves and set-up values of the computing waves, none of it inside the MFMA loop, no scratch.
            # Everything else must not spill at all.
            # (the LDS-DMA staged workgroup split-K parks pass-loop scalars of its K-tail builds the same way, outside the k loop)
            allowed = 40 if (m.group(1) == "SGPRs Spill" and ("persistent" in (name or "") or "streamk" in (name or ""))) else 0
            if m.group(1) == "SGPRs Spill" and "wskd_kernel" in (name or ""):
                allowed = 16
            # (the grouped kernel holds nine unrolled tile loops -- 0..4 m-tiles x lone / shared -- in one function: tile-list, fill-tile
            #  and descriptor scalars of the loops that are not running sit in VGPR lanes; its MFMA blocks read back at most two per k block)
            if m.group(1) == "SGPRs Spill" and "bf16x_grouped_kernel" in (name or ""):
                allowed = 160
            if m.group(1) == "SGPRs Spill" and "bf16x_streamk_kernel" in (name or ""):
                allowed = 64
            assert int(m.group(2)) <= allowed, f"{name}: {m.group(1)} = {m.group(2)}"
        m = re.search(r"VGPRs: (\d+)", line)
        if m and tile_kernel in (name or ""):
            assert int(m.group(1)) <= 256, name
        m = re.search(r"AGPRs: (\d+)", line)
        # (the both-operand image builds read their fragments straight into AGPRs on purpose: MFMA operands, never promoted values)
        # (... and the four-wave builds of dga_launch_menu_j.hip accumulate inside the MFMA: no vector instruction reads an accumulator)
        if m and "gemm_" in (name or "") and "bf16x_image_kernel" not in (name or "") and "gemm_b16_w4_kernel" not in (name or "") and \
                unit != "dga_launch_menu_j.hip":
            assert int(m.group(1)) == 0, f"{name}: MFMA results in AGPRs (a v_accvgpr_read per promoted value in the main loop)"
    assert seen >= min_kernels, seen


def test_the_rccl_host_compiles_and_links():
    """tests/host/sharded_host_rccl.cpp -- INTEGRATION.md section 6 with its RCCL callback (ncclGroupStart / ncclSend / ncclRecv per peer /
    ncclGroupEnd on the stream the executor names, ncclCommInitRank per process) -- compiles against rccl.h and links against
    librccl.so and the in-tree libdga_hip.so on this box; running it needs two GPUs (tests/test_sharded_host_rccl_gpu.py)."""
    lib = ROOT / "deepgemm_ascend_amd" / "libdga_hip.so"
    ora = ROOT / "oracle" / "libdga_oracle.so"
    if not lib.exists() or not ora.exists():
        pytest.skip("libdga_hip.so / libdga_oracle.so not built")
    out = ROOT / "build" / "host" / "sharded_host_rccl"
    out.parent.mkdir(parents=True, exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", f"-I{ROOT / 'include'}", "-I/opt/rocm/include",
           str(ROOT / "tests" / "host" / "sharded_host_rccl.cpp"), "-o", str(out), f"-L{lib.parent}", "-ldga_hip", f"-L{ora.parent}", "-ldga_oracle",
           "-L/opt/rocm/lib", "-lrccl", "-lpthread", f"-Wl,-rpath,{lib.parent}", f"-Wl,-rpath,{ora.parent}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(out)], capture_output=True, text=True, timeout=60).stdout
    for name in ("ncclCommInitRank", "ncclGetUniqueId", "ncclGroupStart", "ncclSend", "ncclRecv", "ncclGroupEnd", "dga_sharded_forward", "dga_sharded_layout"):
        assert re.search(rf"\bU {name}\b", syms), f"{name} is not a dynamic import of the host program"
    needed = subprocess.run(["readelf", "-d", str(out)], capture_output=True, text=True, timeout=60).stdout
    assert "librccl.so" in needed and "libdga_hip.so" in needed
