"""Static check of the hand-placed LDS waits of the bf16-exact image kernel (csrc/gemm_fp8_bf16x_image_kernel.hpp).

The kernel issues its fragment reads (ds_read_b128 into AGPRs) and image stores as inline asm and places the
`s_waitcnt lgkmcnt(N)` itself from a compile-time schedule; the compiler neither sees the reads nor orders the MFMAs
against the waits (only the sched_barrier does).  This script compiles the translation unit to ISA, walks the main loop of
every instantiation twice (steady state) and checks, instruction by instruction, that no instruction reads a register whose
ds_read is still counted in lgkmcnt -- LDS operations of one wave complete in order, so after `lgkmcnt(N)` everything but
the N youngest has landed -- or whose buffer load is still counted in vmcnt (the A-image build fetches its pieces by inline
asm as well: csrc/gemm_fp8_bf16x_aimage_kernel.hpp).  It also checks that the image stores' data registers are not rewritten by the instruction
right behind the store (the >64-bit store-data hazard the hazard recognizer cannot see inside an asm).

    python scripts/check_bximg_waits.py        # exit code 0 = every instantiation passes
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "deepgemm_ascend_amd" / "csrc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"-I{ROOT / 'include'}", f"-I{CSRC}", "-fno-slp-vectorize", "-mllvm",
         "-amdgpu-mfma-vgpr-form=1", "-mllvm", "-pragma-unroll-threshold=1000000"]


def regs(tok: str) -> set[str]:
    """'a[28:31]' -> {'a28', ..., 'a31'}; 'v187' -> {'v187'}; anything else -> {}"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return {f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([av])(\d+)", tok)
    return {f"{m.group(1)}{m.group(2)}"} if m else set()


def main_loop(lines: list[str]) -> list[str]:
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            a = labels[m.group(1)]
            if sum("v_mfma" in x for x in lines[a:i]) >= 64 and (best is None or i - a > best[1] - best[0]):
                best = (a, i)
    if best is None:
        raise SystemExit("no main loop (>= 64 MFMAs) found")
    return [x.strip() for x in lines[best[0]:best[1]] if x.strip() and not x.strip().startswith((";", "."))]


def check(name: str, body: list[str]) -> int:
    pending: list[set[str]] = []   # in-flight LDS operations, oldest first; a store is an empty set
    vm: list[set[str]] = []        # in-flight vector-memory loads (vmcnt, in order too); an LDS-DMA is an empty set
    errors = 0
    n_mfma = n_wait = 0
    for rep in range(2):
        for idx, ins in enumerate(body):
            op, _, rest = ins.partition(" ")
            args = [a for a in rest.split(",")] if rest else []
            if op.startswith("ds_read"):
                pending.append(regs(args[0]))
            elif op.startswith("ds_write"):
                pending.append(set())
                nxt = body[idx + 1] if idx + 1 < len(body) else ""
                nop, _, nrest = nxt.partition(" ")
                if nop.startswith("v_") and nrest and regs(nrest.split(",")[0]) & regs(args[1]):
                    print(f"{name}: store data rewritten by the next instruction: {ins} ; {nxt}")
                    errors += 1
            elif op.startswith("buffer_load") or op.startswith("global_load"):
                to_lds = " lds" in ins or "_lds_" in op   # LDS-DMA: no destination register
                src = set()
                for a in args if to_lds else args[1:]:
                    src |= regs(a)
                if src & (set().union(*vm, *pending) if (vm or pending) else set()):
                    print(f"{name}: {op} takes an address from a register still in flight: {ins}")
                    errors += 1
                vm.append(set() if to_lds else regs(args[0]))
            elif op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    keep = int(m.group(1))
                    vm = vm[len(vm) - keep:] if keep < len(vm) else vm
                    if keep == 0:
                        vm = []
                m = re.search(r"lgkmcnt\((\d+)\)", rest)
                if m:
                    n_wait += rep
                    keep = int(m.group(1))
                    pending = pending[len(pending) - keep:] if keep < len(pending) else pending
                    if keep == 0:
                        pending = []
            elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                print(f"{name}: scalar load inside the loop (shares lgkmcnt, returns out of order): {ins}")
                errors += 1
            elif op.startswith("v_mfma"):
                n_mfma += rep
                src = set()
                for a in args[1:3]:
                    src |= regs(a)
                inflight = set().union(*pending, *vm) if (pending or vm) else set()
                if src & inflight:
                    print(f"{name}: MFMA reads a fragment still in flight (pass {rep}, loop instr {idx}): {ins}")
                    errors += 1
            elif op.startswith("v_") or op.startswith("buffer_store") or op.startswith("global_store"):
                # any other consumer of a register whose load is still counted (a copy the compiler slipped in front of the wait)
                src = set()
                for a in (args[1:] if op.startswith("v_") else args):
                    src |= regs(a)
                inflight = set().union(*pending, *vm) if (pending or vm) else set()
                if src & inflight:
                    print(f"{name}: {op} reads a register still in flight (pass {rep}, loop instr {idx}): {ins}")
                    errors += 1
            if len(pending) > 15:
                # more than the counter can tell apart: a later wait of N <= 15 is still exact, nothing to flag
                pass
    n_vm = sum(1 for x in body if x.startswith(("buffer_load", "global_load")))
    print(f"{name}: {n_mfma} MFMAs, {n_wait} lgkmcnt waits, {n_vm} vector-memory loads per k block, {errors} problem(s)")
    return errors


def main() -> int:
    with tempfile.TemporaryDirectory() as td:
        out = Path(td) / "f.s"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", *FLAGS, "-x", "hip", "--cuda-device-only", "-S",
                               "-o", str(out), str(CSRC / "dga_launch_menu_f.hip")])
        text = out.read_text()
    errors = 0
    for chunk in text.split(".globl")[1:]:
        name = chunk.split("\n", 1)[0].strip()
        if ("bf16x_image_kernel" not in name and "bf16x_aimage_kernel" not in name) or "v_mfma" not in chunk:
            continue
        errors += check(name, main_loop(chunk.split("\n")))
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
