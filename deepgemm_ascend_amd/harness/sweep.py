"""Tuning sweep -- the counterpart of the reference's grid-search driver
(/root/reference/deep_gemm_ascend/framework/benchmark/benchmark.py): same shape list (:24-44), per-rank slice of
the candidate list (:249-253), jsonl results (:322-332), crash-skip checkpoint written BEFORE each run
(:256-304: a combo found in the checkpoint on restart is the one that crashed -> recorded as time -1 and skipped),
correctness gate before timing (:307-315).  Candidates here are the compiled tile variants x raster groups
(the CDNA4 knobs) instead of the Ascend L1/L0 block counts; timing is hipEvents in-process instead of `msprof op`.
The winner per shape is appended to the tiling cache CSV the operator consults ($DGA_CACHE_FILE_PATH).

  python -m deepgemm_ascend_amd.harness.sweep --out sweep_out [--rank R --num-processes P] [--shapes 4096,4096,4096 ...]
"""
from __future__ import annotations

import argparse
import json
import os
from dataclasses import asdict, dataclass, field
from pathlib import Path

import torch

SHAPE_GROUP = [  # M, N, K  (benchmark.py:24-44)
    [4096, 4096, 4096], [8, 7168, 18432], [8, 18432, 7168], [64, 4096, 7168], [64, 7168, 18432], [64, 18432, 7168],
    [64, 24576, 1536], [64, 32768, 512], [64, 7168, 16384], [128, 4096, 7168], [128, 7168, 18432], [128, 18432, 7168],
    [1024, 4096, 7168], [1024, 18432, 7168], [2048, 4096, 7168], [1279, 5003, 7681], [3511, 6151, 8191], [5119, 6997, 9901],
]
TILES = [(256, 256), (128, 256), (256, 128), (128, 128), (64, 256), (64, 128), (32, 256), (32, 128), (16, 256), (16, 128)]
RASTERS = [1, 2, 4, 8, 16]
ERROR_TOL = 1e-4


@dataclass
class Result:
    idx: int
    M: int
    N: int
    K: int
    time: float
    diff: float
    negative: bool
    parameters: dict = field(default_factory=dict)


THREE_STAGE = {(128, 256), (128, 128), (64, 256), (64, 128), (32, 256), (32, 128), (16, 256), (16, 128)}   # tiles with a 3-stage build
LOADER_WAVES = {(128, 256), (128, 128), (64, 256), (64, 128), (16, 128)}   # ... of which these have a loader-wave variant (dispatchPolicyTag 4)
PINGPONG = {(256, 256)}                               # ... ping-pong / continuous schedules (dispatchPolicyTag 1 / 2)


def heuristic_raster(m, n, bm, bn, splitk=1, xcds=8, stages=2):
    """The raster group select_mi355x derives for a tile (dga_tiling.cpp): the largest power of two whose square fits
    the tiles an XCD runs AT THE SAME TIME (its 32 CUs x workgroups per CU), so that those form a near-square patch."""
    tiles_m = -(-m // bm)
    lds = (max(bm, 32) * 128 + bn * 128 + ((bm + 8 + 255) // 256) * 256 * 4) * (3 if stages == 3 else 2)
    waves = 8 if (bm, bn) == (256, 256) else 4
    wg_per_cu = max(1, min(160 * 1024 // lds, 2048 // (waves * 64)))
    conc = min(max(1, (tiles_m * -(-n // bn) * splitk) // xcds), 32 * wg_per_cu)
    gm = 1
    while (gm * 2) * (gm * 2) <= conc and gm * 2 <= tiles_m:
        gm *= 2
    return min(gm, 255)


def grid_shapes(count, seed=0, max_mnk=2 ** 39):
    """Random shapes for the predictor's training set: M log-uniform in [8, 8192], N and K multiples of 128 in
    [512, 16384] (log-uniform), M*N*K bounded so that one launch stays under a millisecond."""
    import random
    rng = random.Random(seed)
    out = []
    while len(out) < count:
        m = int(round(2 ** rng.uniform(3, 13)))
        if rng.random() < 0.5:
            m = max(8, (m + 15) // 16 * 16)
        n = max(512, int(round(2 ** rng.uniform(9, 14))) // 128 * 128)
        k = max(512, int(round(2 ** rng.uniform(9, 14))) // 128 * 128)
        if m * n * k <= max_mnk and [m, n, k] not in out:
            out.append([m, n, k])
    return out


def candidates(m, n, k, rasters=None):
    out = []
    kb = -(-k // 128)
    for bm, bn in TILES:
        if bm >= 2 * max(m, 16) and bm > 16:      # filter_parameters analogue: tiles twice the problem are pointless
            continue
        blocks = -(-m // bm) * -(-n // bn)
        splits = [1] + [s for s in (2, 3, 4, 5, 6, 8, 16) if blocks * s <= 1024 and kb // s >= 4 and blocks < 192]
        for r in (RASTERS if rasters is None else [0]):
            if r > max(1, -(-m // bm)):
                continue
            for st in ([2, 3] if (bm, bn) in THREE_STAGE else [2]):
                for sk in splits:
                    # 3-stage builds: the plain loop and its loader-wave variant (dispatchPolicyTag 4)
                    # ... and, where the raster holds more tiles than the chip has CUs, the persistent form (5)
                    lw = ([0, 4, 5] if sk == 1 and blocks > CUS else [0, 4]) if st == 3 and (bm, bn) in LOADER_WAVES else [0]
                    # 256x256: the three schedules and, on rasters of full tiles with more tiles than CUs, the persistent
                    # continuous pipeline (6)
                    full = m % 256 == 0 and n % 256 == 0 and k % 128 == 0 and k >= 256 and blocks > CUS
                    for pol in (([0, 1, 2, 6] if full else [0, 1, 2]) if (bm, bn) in PINGPONG and sk == 1 else lw):
                        rr = r if rasters is None else heuristic_raster(m, n, bm, bn, sk, stages=st)
                        out.append({"m1": bm, "n1": bn, "raster": rr, "stages": st, "splitk": sk, "policy": pol})
                        # 256x256, more than one wave of tiles with a small remainder: also with the quarter-tile tail
                        if (bm, bn) == (256, 256) and sk == 1 and blocks > 256 and 0 < blocks % 256 <= 64:
                            out.append({"m1": bm, "n1": bn, "raster": rr, "stages": st, "splitk": sk, "policy": pol, "tail": 1})
    # short-M: the one-launch workgroup split-K kernel (csrc/gemm_fp8_wsk_kernel.hpp, kernelSerial 6): 8 waves = 8 K slices
    if m <= 64 and k % 16 == 0 and k > 0:
        # (wsk names the build: 1 = fragments global -> registers (tiling.build = 1; the candidate record keeps the "stages": 1 it
        #  had when the name rode on that field), 2 = per-wave LDS-DMA rings, M <= 32)
        out.append({"m1": 16 if m <= 16 else (32 if m <= 32 else 64), "n1": 128, "raster": 1, "stages": 1, "splitk": 1, "policy": 0, "wsk": 1})
        if m <= 32:
            out.append({"m1": 16 if m <= 16 else 32, "n1": 128, "raster": 1, "stages": 3, "splitk": 1, "policy": 0, "wsk": 2})
    return out


BX_TILES = [(128, 256), (128, 128), (64, 256), (64, 128), (32, 128)]    # the bf16-exact policy's menu (csrc/dga_launch_menu_e.hip)


def bx_raster(m, n, bm, bn, cus=256):
    """The raster group dga_tiling_bf16_exact gives a tile: the XCD's concurrent patch square in operand rows."""
    tiles_m = -(-m // bm)
    tiles = tiles_m * -(-n // bn)
    wpc = 2 if bm * bn <= 64 * 128 else 1
    conc = min(max(1, tiles // 8), (cus // 8) * wpc)
    gm = 1
    while (gm * 2) * (gm * 2) * bm <= conc * bn and gm * 2 <= tiles_m:
        gm *= 2
    return gm


def candidates_bx(m, n, k, cus=256):
    """Candidates of the bf16-exact policy (dispatchPolicyTag 7) for a dense problem: its five tiles x split-K, and for the 128 x 256
    tile the quarter-tile tail (kernelSerial 5), the one-launch Stream-K (7) and the one-tile build beside the persistent one; the
    workgroup split-K (6) for decode rows.  The reference's grid under the policy's own constraints
    (/root/reference/get_best_config/catlass_parameter.py:68-126; driver benchmark.py:47-193)."""
    kb = -(-k // 128)
    out = []
    for bm, bn in BX_TILES:
        if bm > 32 and bm >= 4 * max(m, 16):      # a tile four times the rows there are
            continue
        tiles = -(-m // bm) * -(-n // bn)
        rr = bx_raster(m, n, bm, bn, cus)
        splits = [1] + [sk for sk in (2, 3, 4, 6, 8, 16) if kb // sk >= 4 and tiles * sk <= 4 * cus and tiles < cus and sk * m * n * 4 <= (512 << 20)]
        for sk in splits:
            out.append({"m1": bm, "n1": bn, "raster": rr, "stages": 3, "splitk": sk, "policy": 7})
        if (bm, bn) == (128, 256):
            if tiles > cus:
                out.append({"m1": bm, "n1": bn, "raster": rr, "stages": 3, "splitk": 1, "policy": 7, "build": 8})     # one tile per workgroup
                if 0 < tiles % cus <= cus // 2:
                    out.append({"m1": bm, "n1": bn, "raster": rr, "stages": 3, "splitk": 1, "policy": 7, "tail": 1})
            if tiles % cus and kb >= 4:
                out.append({"m1": bm, "n1": bn, "raster": rr, "stages": 3, "splitk": 1, "policy": 7, "streamk": 1})
    if m <= 32 and k % 16 == 0 and kb >= 8:
        out.append({"m1": 16 if m <= 16 else 32, "n1": 128, "raster": 1, "stages": 3, "splitk": 1, "policy": 7, "wsk": 2})
    # the one-launch split-K of the 64 x 128 tile (build 10, csrc/gemm_fp8_bf16x_dsk_kernel.hpp): every workgroup resident at once
    dt = -(-m // 64) * -(-n // 128)
    if 16 < m <= 512 and k % 16 == 0 and kb >= 4 and dt <= cus:
        smax = max(1, min(8, cus // dt, kb // 4))
        for sk in sorted({smax, max(1, (3 * smax) // 4), max(1, smax // 2)}):
            out.append({"m1": 64, "n1": 128, "raster": 1, "stages": 3, "splitk": sk, "policy": 7, "wsk": 2, "build": 10})
    return out


def bx_serial(p):
    return 6 if p.get("wsk") else (7 if p.get("streamk") else (5 if p.get("tail") else (4 if p["splitk"] > 1 else 0)))


# ---- the compiled menu (csrc/dga_launch.hip kVariants) and its constraints -------------------------------------------
# (bm, bn, wavesM, wavesN, stages, dispatch policies the build exists in)
MENU = [(256, 256, 4, 2, 2, (0, 1, 2, 6)), (128, 256, 2, 2, 2, (0, 2)), (256, 128, 4, 1, 2, (0, 2)), (128, 128, 2, 2, 2, (0, 2)),
        (64, 256, 1, 4, 2, (0, 2)), (64, 128, 1, 4, 2, (0,)), (128, 256, 2, 4, 2, (0, 2)), (128, 256, 2, 4, 3, (0,)),
        (128, 256, 2, 2, 3, (0, 4, 5)), (128, 128, 2, 2, 3, (0, 4, 5)), (64, 256, 1, 4, 3, (0, 4, 5)), (32, 256, 1, 4, 2, (0,)),
        (32, 128, 1, 4, 2, (0,)), (16, 256, 1, 4, 2, (0,)), (16, 128, 1, 4, 2, (0,)),
        (64, 128, 1, 4, 3, (0, 4, 5)), (32, 256, 1, 4, 3, (0,)), (32, 128, 1, 4, 3, (0,)), (16, 256, 1, 4, 3, (0,)), (16, 128, 1, 4, 3, (0, 4, 5))]
LDS_BYTES, ACC_REGS, CUS = 160 * 1024, 128, 256


def stage_bytes(bm, bn, waves):
    """GemmCfg::STAGE_BYTES (csrc/dga_device_common.hpp): A rows padded to whole DMA instructions, scale slots."""
    dnt = waves * 64
    a_rows = bm if bm * 8 >= dnt else dnt // 8
    return a_rows * 128 + bn * 128 + (-(-(bm + 8) // dnt) * dnt) * 4


def check_candidate(prob, c):
    """Per-variant constraint checker -- the counterpart of the reference's filter_parameters / per-kernel checkers
    (/root/reference/get_best_config/catlass_parameter.py:308-368: L1 / L0C capacity, smallmatmul needs one K step and one
    tile per core, split-K needs a long K ...).  prob: dict(m, n, k, groups, layout in {"dense","masked","contiguous"},
    rows_per_group).  Returns (ok, reason)."""
    bm, bn, wm, wn, st, pol, sk = c["m1"], c["n1"], c["wavesM"], c["wavesN"], c["stages"], c["policy"], c["splitk"]
    build = [v for v in MENU if v[:5] == (bm, bn, wm, wn, st)]
    if not build:
        return False, "no such build in the menu"
    if pol not in build[0][5]:
        return False, f"dispatch policy {pol} not compiled for this build"
    if stage_bytes(bm, bn, wm * wn) * st > LDS_BYTES:                       # JudgeSpace: L1 (here LDS) capacity
        return False, "LDS"
    if bm * bn // (wm * wn * 64) > ACC_REGS:                               # JudgeSpace: L0C (here accumulator VGPRs)
        return False, "accumulator registers"
    kb = -(-prob["k"] // 128)
    if pol == 5:   # persistent loader waves: whole rasters with more tiles than CUs (otherwise it IS policy 4)
        groups = prob["groups"] if prob["layout"] == "masked" else 1
        # (a masked grouped problem is a weight stream: the persistent build is also the one that moves weights and outputs
        #  with the non-temporal policy, whatever the tile count)
        if sk != 1 or (prob["layout"] != "masked" and groups * -(-prob["m"] // bm) * -(-prob["n"] // bn) <= CUS):
            return False, "persistent form: more tiles than CUs, no split-K"
    if pol == 6 and (prob["layout"] != "dense" or prob["m"] % 256 or prob["n"] % 256 or prob["k"] % 128 or prob["k"] < 256 or
                     sk != 1 or (prob["m"] // 256) * (prob["n"] // 256) <= CUS):
        return False, "persistent continuous pipeline: dense rasters of full 256x256 tiles, more tiles than CUs"
    if prob["layout"] == "masked":
        need = 16 if prob["m"] <= 16 else 32 if prob["m"] <= 32 else 64 if prob["m"] <= 64 else 128
        if prob["m"] <= 128 and bm != need:
            return False, "masked layout: one tile row per expert (B is read once)"
        if sk != 1:
            return False, "split-K is dense only"
    elif prob["layout"] == "contiguous":
        tall = bm == 256 and bn == 256 and prob["rows_per_group"] >= 512
        if not tall and (bm > 128 or 128 % bm):
            return False, "contiguous layout: a tile must not straddle two group segments"
        if sk != 1:
            return False, "split-K is dense only"
    else:
        blocks = -(-prob["m"] // bm) * -(-prob["n"] // bn)
        if bm >= 2 * max(prob["m"], 16) and bm > 16:
            return False, "tile twice the problem"
        if sk > 1 and not (blocks * sk <= 1024 and kb // sk >= 4 and blocks < 192 and pol == 0):
            return False, "split-K needs few tiles, >= 4 k blocks per split and the plain loop"
    return True, ""


def grouped_candidates(prob):
    out = []
    for (bm, bn, wm, wn, st, pols) in MENU:
        for pol in pols:
            if pol == 1:
                continue
            c = {"m1": bm, "n1": bn, "wavesM": wm, "wavesN": wn, "stages": st, "policy": pol, "splitk": 1, "raster": 0}
            if check_candidate(prob, c)[0]:
                out.append(c)
    return out


def gen_data(m, n, k, seed=0):
    """benchmark.py:343-367 analogue, fp8: N(0,1) data, amax block scaling, e4m3fn codes; golden = fp32 matmul of the
    dequantised operands on the device (TF32 is not a thing on gfx950: this is exact-fp32 MFMA / VALU)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    kp, np_ = -(-k // 128) * 128, -(-n // 128) * 128
    xa = torch.zeros((m, kp), device="cuda"); xa[:, :k] = torch.randn((m, k), device="cuda", generator=g)
    xb = torch.zeros((np_, kp), device="cuda"); xb[:n, :k] = torch.randn((n, k), device="cuda", generator=g)
    sa = xa.view(m, kp // 128, 128).abs().amax(2).clamp_min(1e-30) / 448
    qa = (xa.view(m, kp // 128, 128) / sa[..., None]).reshape(m, kp).to(torch.float8_e4m3fn)
    sb = xb.view(np_ // 128, 128, kp // 128, 128).abs().amax((1, 3)).clamp_min(1e-30) / 448
    qb = (xb.view(np_ // 128, 128, kp // 128, 128) / sb[:, None, :, None]).reshape(np_, kp).to(torch.float8_e4m3fn)
    da = (qa.float().view(m, kp // 128, 128) * sa[..., None]).reshape(m, kp)[:, :k]
    dbm = (qb.float().view(np_ // 128, 128, kp // 128, 128) * sb[:, None, :, None]).reshape(np_, kp)[:n, :k]
    golden = da @ dbm.T
    s_abs = da.abs() @ dbm.abs().T                 # S of the parity bar (harness/tolerance.py)
    a = qa.view(torch.uint8)[:, :k].contiguous(); b = qb.view(torch.uint8)[:n, :k].contiguous()
    return a, sa.contiguous(), b, sb.contiguous(), golden, s_abs


def is_correct(golden, out, s_abs=None, policy="fast", short_k=False):
    """The correctness gate in front of every timing (benchmark.py:384-398's role), on the product's ONE parity bar
    (harness/tolerance.py): every element within 2 ulp_bf16 + eps(policy) * S of the bf16-rounded golden.  s_abs = S, the
    |a| |b| product of the dequantised operands (gen_data returns it); returns (ok, fraction of elements beyond 2 ulp)."""
    from . import tolerance
    if s_abs is None:
        raise ValueError("is_correct needs S (gen_data's sixth return value): the bar is S-based")
    ok, rep = tolerance.check(out.float(), tolerance.bf16_round(golden), s_abs, policy=policy, short_k=short_k, golden_order="any")
    return ok, rep["frac_gt_2ulp"]


_TIMING_STREAM = []


def graph_us(fn, iters, replays=3, prewarm_ms=0.0):
    """Device time per call of `fn`: `iters` calls captured into one HIP graph on the harness's side stream, replayed `replays`
    times between two events.  Issued one by one from Python a call costs 8-12 us of host time, which hides every difference
    between candidates whose kernels are shorter (profiles/r03_host_overhead.txt).  None where the capture fails."""
    if not _TIMING_STREAM:
        _TIMING_STREAM.append(torch.cuda.Stream())
    side = _TIMING_STREAM[0]
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            fn()                      # the capture stream's workspace exists before the capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    except Exception:
        torch.cuda.synchronize()
        return None
    import time as _time
    g.replay()
    torch.cuda.synchronize()
    t0 = _time.perf_counter()
    while (_time.perf_counter() - t0) * 1e3 < prewarm_ms:
        g.replay()
        torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (iters * replays)
    del g
    return us


def time_us(fn, warm=3, iters=10, device_time=False):
    """Launch interval of `iters` back-to-back calls (HIP events); device_time=True: by graph replay (graph_us), so that calls
    shorter than their own host cost are told apart."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    if device_time:
        us = graph_us(fn, iters)
        if us is not None:
            return us
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


COLD_MAX_M = 256          # --cold: shapes this short are decode GEMMs -- their weights come from HBM on every call
INFINITY_CACHE = 256 << 20


def benchmark_shape(shape, out_dir: Path, rank=0, num_processes=1, iters=10, rasters=None, prewarm_s=0.15, only_policy=None,
                    cold=False, versus_tuned=False, arith="fast"):
    """cold: time the candidates of a short-M shape on operand sets rotated past the Infinity Cache (SURVEY.md 8(d)'s
    protocol).  Re-launching on one set keeps a decode shape's 20-140 MB of weights in the 256 MiB cache, which favours
    tilings that leave CUs idle (fewer, longer streams): 64x4096x7168 picks 16x128 without split-K warm (16.2 us) and pays
    26.9 us for it cold, where a 4-way split takes 17.6 us either way (scripts/decode_cold_sweep.py)."""
    import deepgemm_ascend_amd as dga
    m, n, k = shape
    bx = arith == "bf16_exact"     # the in-contract policy's own menu (candidates_bx), gated on ITS bar, same checkpoint / jsonl protocol
    cands = candidates_bx(m, n, k) if bx else candidates(m, n, k, rasters)
    if versus_tuned and not bx:   # a supplementary sweep of the kernels outside the tile menu against the operator's current pick
        t0 = dga.tiling(m, n, k)
        pick = {"m1": int(t0.m1), "n1": int(t0.n1), "raster": int(t0.swizzleOffset), "stages": 1 if t0.build == 1 else int(t0.stages),
                "splitk": int(t0.splitkFactor), "policy": int(t0.dispatchPolicyTag)}
        if t0.kernelSerial == 6:
            pick["wsk"] = 2 if (t0.build != 1 and m <= 32) else 1
        elif t0.kernelSerial == 5:
            pick["tail"] = 1
        cands = [pick] + [c for c in cands if c.get("wsk") and c != pick]
    if only_policy is not None:   # a supplementary sweep of one dispatch policy (merged into an earlier run's records)
        cands = [c for c in cands if c["policy"] == only_policy]
        if not cands:
            return None
    per = -(-len(cands) // num_processes)
    lo, hi = rank * per, min(len(cands), (rank + 1) * per)
    tagx = "bx_" if bx else ""
    res_path = out_dir / f"shape_{tagx}{m}_{n}_{k}_rank_{rank}.jsonl"
    ck_path = out_dir / f"shape_{tagx}{m}_{n}_{k}_rank_{rank}_checkpoint.jsonl"
    last = -1
    if ck_path.exists():
        try:
            last = json.loads(ck_path.read_text().strip().splitlines()[-1])["last_process_idx"]
        except Exception:
            last = -1
    a, sfa, b, sfb, golden, s_abs = gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    # clock pre-warm: after the idle gap of the data generation the GPU needs ~100 ms of work to reach its sustained
    # clocks; without it the first candidates of every shape (the 256x256 builds) are timed 10-15 % slow
    import time as _time
    t_warm = dga.tiling(m, n, k, policy="bf16_exact" if bx else None)
    t0 = _time.perf_counter()
    while _time.perf_counter() - t0 < prewarm_s:
        for _ in range(20):
            dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t_warm, policy="bf16_exact" if bx else "fast")
        torch.cuda.synchronize()
    opbytes = m * k + n * k + 2 * m * n
    sets = [(a, sfa, b, sfb, out)]
    if cold and m <= COLD_MAX_M and opbytes < INFINITY_CACHE:
        for _ in range(min(16, max(3, -(-(320 << 20) // opbytes))) - 1):
            sets.append((a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty_like(out)))
    turn = [0]
    best = None
    for idx in range(lo, hi):
        if idx < last:
            continue
        p = cands[idx]
        if idx == last:   # the previous process died while running this combo: record and skip
            with open(res_path, "a") as f:
                f.write(json.dumps(asdict(Result(idx, m, n, k, -1, -1, True, p))) + "\n")
            continue
        ck_path.write_text(json.dumps({"last_process_idx": idx}) + "\n")
        t = dga.tiling(m, n, k, policy="bf16_exact" if bx else None)
        t.m1, t.n1, t.swizzleOffset = p["m1"], p["n1"], p["raster"]
        t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag = (3 if p["stages"] == 1 else p["stages"]), 0, 0, p["policy"]
        t.build = p.get("build", 0) if bx else (1 if p.get("wsk") == 1 else 0)
        t.splitkFactor = p["splitk"]
        t.kernelSerial = bx_serial(p) if bx else (6 if p.get("wsk") else (5 if p.get("tail") else (4 if p["splitk"] > 1 else 0)))
        if dga.tiling_check(t) != 0:     # (a candidate the compiled menu does not hold: recorded, not run)
            with open(res_path, "a") as f:
                f.write(json.dumps(asdict(Result(idx, m, n, k, -1, -1, True, dict(p, refused=1)))) + "\n")
            continue
        def fn():
            c = sets[turn[0] % len(sets)]
            turn[0] += 1
            dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t, policy="bf16_exact" if bx else None)
        turn[0] = 0
        fn(); torch.cuda.synchronize()
        ok, diff = is_correct(golden, out, s_abs, policy="bf16_exact" if bx else "fast", short_k=k < 128)
        n_it = -(-max(iters, 2 * len(sets)) // len(sets)) * len(sets)   # whole turns of the operand sets (a graph replays them in order)
        turn[0] = 0
        if ok and bx:   # (behind a clock pre-warm, five replays: the policy's candidates differ by a few per cent)
            for _ in range(max(3, len(sets))):
                fn()
            torch.cuda.synchronize()
            us = graph_us(fn, n_it, replays=5, prewarm_ms=25.0) or time_us(fn, warm=3, iters=n_it)
        else:
            us = time_us(fn, warm=max(3, len(sets)), iters=n_it, device_time=True) if ok else 999999999
        if len(sets) > 1:
            p = dict(p, cold_sets=len(sets))
        with open(res_path, "a") as f:
            f.write(json.dumps(asdict(Result(idx, m, n, k, us, diff, not ok, p))) + "\n")
        if ok and (best is None or us < best[0]):
            best = (us, p)
    ck_path.write_text(json.dumps({"last_process_idx": hi}) + "\n")
    return best


GROUPED_SHAPES = (   # (layout, groups, rows per group (m_max), n, k): DeepSeek-V3-like MoE layers and their decode-time masks
    [("masked", g, mm, n, k) for (n, k) in ((2048, 7168), (7168, 2048), (4096, 7168)) for g in (32, 256) for mm in (16, 64, 128)] +
    [("contiguous", g, r, n, k) for (n, k) in ((4096, 7168), (7168, 2048)) for (g, r) in ((32, 128), (8, 1024), (4, 4096))])


def benchmark_grouped(shape, out_dir: Path, iters=10, prewarm_s=0.15):
    """One masked / contiguous grouped problem: every legal build timed (full mask), gate = bit pattern of two groups
    against the strict kernel under the fast path's bar; records go to the same jsonl format as the dense sweep."""
    import time as _time
    import deepgemm_ascend_amd as dga
    layout, groups, rows, n, k = shape
    gen = torch.Generator(device="cuda").manual_seed(0)
    rf = lambda sh: (lambda x: torch.where((x & 0x7F) == 0x7F, x & 0x80, x))(torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=gen))
    kb, nb = k // 128, n // 128
    b = rf((groups, n, k)); sfb = torch.rand((groups, nb, kb), device="cuda", generator=gen) + 0.5
    if layout == "masked":
        prob = {"m": rows, "n": n, "k": k, "groups": groups, "layout": layout, "rows_per_group": rows}
        a = rf((groups, rows, k)); sfa = torch.rand((groups, rows, kb), device="cuda", generator=gen) + 0.5
        out = torch.empty((groups, rows, n), dtype=torch.bfloat16, device="cuda")
        masked = torch.full((groups,), rows, dtype=torch.int32, device="cuda")
        base = lambda: dga.tiling(rows, n, k, groups=groups, expected_m=rows)
        run = lambda t, strict=False: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, rows, tiling_=t, strict=strict)
    else:
        msum = groups * rows
        prob = {"m": msum, "n": n, "k": k, "groups": groups, "layout": layout, "rows_per_group": rows}
        a = rf((msum, k)); sfa = torch.rand((msum, kb), device="cuda", generator=gen) + 0.5
        out = torch.empty((msum, n), dtype=torch.bfloat16, device="cuda")
        idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(rows).contiguous()
        base = lambda: dga.tiling(msum, n, k, groups=groups, contiguous=True)
        run = lambda t, strict=False: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t, strict=strict)
    from . import tolerance
    run(base(), strict=True); torch.cuda.synchronize()
    golden = out.float().clone()
    # S of the parity bar: the strict kernel on |a|, |b|, |scales| (bf16-rounded: 2^-9 relative)
    a_keep, b_keep = a.clone(), b.clone()
    a &= 0x7F; b &= 0x7F
    run(base(), strict=True); torch.cuda.synchronize()
    s_abs = out.float().clone()
    a.copy_(a_keep); b.copy_(b_keep)
    del a_keep, b_keep
    t0 = _time.perf_counter()
    while _time.perf_counter() - t0 < prewarm_s:
        run(base()); torch.cuda.synchronize()
    tag = f"{layout}_{groups}x{rows}_{n}_{k}"
    res_path = out_dir / f"shape_{tag}_rank_0.jsonl"
    best = None
    for i, c in enumerate(grouped_candidates(prob)):
        t = base()
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = c["m1"], c["n1"], c["wavesM"], c["wavesN"], c["stages"], c["policy"]
        t.splitkFactor, t.kernelSerial = 1, 0
        tiles = -(-prob["m"] // c["m1"]) * -(-n // c["n1"])
        t.blockDim = tiles * (groups if layout == "masked" else (2 if c["m1"] == 256 else 1))
        c = dict(c, raster=int(t.swizzleOffset), groups=groups, layout=layout, rows_per_group=rows)
        try:
            run(t); torch.cuda.synchronize()
            # uniformly random bytes: the fast path's arbitrary-bit-pattern envelope (tolerance.py, short_k)
            ok, rep = tolerance.check(out.float(), golden, s_abs, policy="fast", short_k=True)
            ratio = rep["frac_gt_2ulp"]
            us = time_us(lambda: run(t), iters=iters) if ok else 999999999
        except Exception:   # a tiling the launcher refuses (recorded, not fatal)
            ok, ratio, us = False, -1.0, 999999999
        with open(res_path, "a") as f:
            f.write(json.dumps(asdict(Result(i, prob["m"], n, k, us, ratio, not ok, c))) + "\n")
        if ok and (best is None or us < best[0]):
            best = (us, c)
    return prob, best


GROUPED_CSV_HEAD = ("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
                    "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag,groups,contiguous\n")
FULL_CSV_HEAD = GROUPED_CSV_HEAD.strip() + ",build\n"      # + dga_tiling_t.build (ABI 7)


def bx_row(m, n, k, p, cus=256):
    """The cache row of a bf16-exact winner: dispatchPolicyTag 7 keys it into that policy's class (csrc/dga_tiling.cpp Cache)."""
    tiles = -(-m // p["m1"]) * -(-n // p["n1"])
    serial = bx_serial(p)
    block_dim = tiles * p["splitk"]
    if serial == 5:
        block_dim = tiles - tiles % cus + 4 * (tiles % cus)
    elif serial == 7:
        block_dim = cus
    elif serial == 6 and p.get("build") != 10:
        block_dim = min(-(-n // 16), cus)
    return (f"{m},{n},{k},{p['m1']},{p['n1']},128,{serial},0,0,0,{block_dim},{p['splitk']},3,{p['raster']},0,0,7,1,0,{p.get('build', 0)}\n")


def write_bx_rows(path, winners):
    """Append tag-7 rows to a cache CSV that has (or gets) the full header; rows of the same shape and class already there are replaced."""
    path = Path(path)
    lines = path.read_text().splitlines(keepends=True) if path.exists() else []
    if not lines:
        lines = [FULL_CSV_HEAD]
    head = lines[0].strip().split(",")
    if "build" not in head:      # an older file: give every row the column
        assert head == GROUPED_CSV_HEAD.strip().split(","), "the cache file must carry the CDNA4 and group columns"
        lines = [FULL_CSV_HEAD] + [ln.rstrip("\n") + ",0\n" for ln in lines[1:] if ln.strip()]
    keys = {(m, n, k) for (m, n, k), _ in winners}
    def is_old_bx(ln):
        c = ln.strip().split(",")
        return len(c) >= 19 and (int(c[0]), int(c[1]), int(c[2])) in keys and (int(c[16]) & 15) == 7 and int(c[17]) <= 1 and int(c[18]) == 0
    lines = [lines[0]] + [ln for ln in lines[1:] if ln.strip() and not is_old_bx(ln)]
    lines += [bx_row(m, n, k, p) for (m, n, k), p in winners]
    path.write_text("".join(lines))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="sweep_out")
    ap.add_argument("--rank", type=int, default=int(os.environ.get("RANK", 0)))
    ap.add_argument("--num-processes", type=int, default=int(os.environ.get("WORLD_SIZE", 1)))
    ap.add_argument("--shapes", nargs="*", default=None, help="M,N,K ...")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cache-csv", default=None, help="append winners to this tiling-cache CSV")
    ap.add_argument("--grid", type=int, default=0, help="sweep this many random shapes (predictor training set)")
    ap.add_argument("--grid-seed", type=int, default=0)
    ap.add_argument("--prewarm-ms", type=float, default=150.0, help="clock pre-warm in front of every shape's candidates")
    ap.add_argument("--heuristic-raster", action="store_true",
                    help="one raster per candidate (the heuristic's) instead of the raster sweep")
    ap.add_argument("--only-policy", type=int, default=None, help="time only the candidates of this dispatchPolicyTag")
    ap.add_argument("--versus-tuned", action="store_true",
                    help="time only the operator's current pick and the workgroup split-K builds (a supplementary sweep)")
    ap.add_argument("--cold", action="store_true",
                    help=f"shapes with M <= {COLD_MAX_M}: rotate operand sets past the Infinity Cache (decode weights are never warm)")
    ap.add_argument("--arith", default="fast", choices=["fast", "bf16_exact"],
                    help="the arithmetic policy whose menu is swept: the fast path's (dispatchPolicyTag 0-6) or the in-contract "
                         "bf16-exact policy's (7: the operator's default; winners become tag-7 rows of the cache CSV)")
    ap.add_argument("--grouped", action="store_true",
                    help="sweep the masked / contiguous grouped shapes (GROUPED_SHAPES) instead of the dense list")
    ap.add_argument("--grouped-shapes", nargs="*", default=None,
                    help="with --grouped: layout,groups,rows,n,k ... instead of GROUPED_SHAPES (layout: masked | contiguous)")
    a = ap.parse_args(argv)
    torch.cuda.set_device(a.rank % max(1, torch.cuda.device_count()))
    out_dir = Path(a.out); out_dir.mkdir(parents=True, exist_ok=True)
    if a.grouped:
        rows_out = []
        gshapes = GROUPED_SHAPES
        if a.grouped_shapes:
            gshapes = [(s.split(",")[0],) + tuple(int(x) for x in s.split(",")[1:]) for s in a.grouped_shapes]
        for shape in gshapes[a.rank::a.num_processes]:
            prob, best = benchmark_grouped(shape, out_dir, a.iters, a.prewarm_ms / 1e3)
            if best:
                us, c = best
                print(json.dumps({"shape": list(shape), "best_us": round(us, 2), **c}), flush=True)
                tiles = -(-prob["m"] // c["m1"]) * -(-prob["n"] // c["n1"])
                contiguous = 1 if prob["layout"] == "contiguous" else 0
                block_dim = tiles * (1 if contiguous else prob["groups"]) * (2 if contiguous and c["m1"] == 256 else 1)
                rows_out.append(f"{prob['m']},{prob['n']},{prob['k']},{c['m1']},{c['n1']},128,0,0,0,0,{block_dim},1,{c['stages']},"
                                f"{c['raster']},{c['wavesM']},{c['wavesN']},{c['policy']},{prob['groups']},{contiguous}\n")
        if a.cache_csv and rows_out:
            new = not Path(a.cache_csv).exists()
            with open(a.cache_csv, "a") as f:
                if new:
                    f.write(GROUPED_CSV_HEAD)
                f.writelines(rows_out)
        return
    shapes = [[int(x) for x in s.split(",")] for s in a.shapes] if a.shapes else SHAPE_GROUP
    if a.grid:
        shapes = grid_shapes(a.grid, a.grid_seed)
    winners = []
    for shape in shapes:
        best = benchmark_shape(shape, out_dir, a.rank, a.num_processes, a.iters, [0] if a.heuristic_raster else None,
                               a.prewarm_ms / 1e3, a.only_policy, a.cold, a.versus_tuned, a.arith)
        if best:
            us, p = best
            m, n, k = shape
            print(json.dumps({"shape": shape, "best_us": round(us, 2), "tflops": round(2.0 * m * n * k / us / 1e6, 1), **p}), flush=True)
            winners.append((shape, p))
    if a.cache_csv and winners and a.arith == "bf16_exact":
        write_bx_rows(a.cache_csv, winners)
    elif a.cache_csv and winners:
        new = not Path(a.cache_csv).exists()
        with open(a.cache_csv, "a") as f:
            if new:   # the reference's 11 columns (csv.cpp:23-26) + the CDNA4 columns the cache reads when present
                f.write("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
                        "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag\n")
            for (m, n, k), p in winners:
                blocks = -(-m // p["m1"]) * -(-n // p["n1"]) * p["splitk"]
                if p.get("tail"):   # whole waves + the last partial wave in quarter tiles
                    blocks = blocks - blocks % 256 + 4 * (blocks % 256)
                f.write(f"{m},{n},{k},{p['m1']},{p['n1']},128,{6 if p.get('wsk') else (5 if p.get('tail') else (4 if p['splitk'] > 1 else 0))},0,0,0,{blocks},"
                        f"{p['splitk']},{p['stages']},{p['raster']},0,0,{p['policy']}\n")


if __name__ == "__main__":
    main()
