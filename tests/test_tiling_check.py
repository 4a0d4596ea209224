"""CPU: dga_tiling_check -- a caller-written dga_tiling_t is data from outside; what the compiled menu does not hold is refused
before any launch (every fp8 GEMM entry calls the same check first; the GPU half is tests/test_tiling_check_gpu.py).
Counterpart of CatlassDynamicMatmulTilingFunc returning GRAPH_FAILED on what it cannot tile
(/root/reference/aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul_tiling.cpp:86-100)."""
import ctypes
import itertools
import random

import pytest

import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import _lib

OK, E_TILING, E_RANGE = 0, -6, -9
FAST_TILES = {(256, 256), (128, 256), (256, 128), (128, 128), (64, 256), (64, 128), (32, 256), (32, 128), (16, 256), (16, 128)}


def _copy(t):
    c = _lib.Tiling()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(t), ctypes.sizeof(_lib.Tiling))
    return c


def test_every_tiling_the_selectors_write_passes():
    n = 0
    for m, nn, k in itertools.product((1, 16, 64, 100, 128, 512, 1279, 4096), (128, 2048, 4096, 5003, 18432), (128, 2048, 7168, 7681)):
        for pol in (None, "bf16_exact", "strict"):
            assert dga.tiling_check(dga.tiling(m, nn, k, policy=pol)) == OK, (m, nn, k, pol)
            n += 1
    for g, em in ((256, 128), (256, 16), (32, 64), (8, 4)):
        for pol in (None, "bf16_exact"):
            assert dga.tiling_check(dga.tiling(128, 2048, 7168, groups=g, expected_m=em, policy=pol)) == OK
    assert dga.tiling_check(dga.tiling(8192, 4096, 7168, groups=8, contiguous=True)) == OK
    assert n > 400


def test_fields_outside_the_menu_are_refused():
    base = dga.tiling(4096, 4096, 4096)
    assert (base.m1, base.n1) == (256, 256) and dga.tiling_check(base) == OK
    for field, bad in (("kernelSerial", 3), ("kernelSerial", 8), ("kernelSerial", 255), ("dispatchPolicyTag", 8), ("dispatchPolicyTag", 32),
                       ("dispatchPolicyTag", 0x80 | 2), ("k1", 64), ("k1", 256), ("m1", 0), ("n1", 0), ("m1", 512), ("m1", 48), ("n1", 64),
                       ("stages", 1), ("stages", 4), ("stages", 9), ("wavesM", 3), ("wavesN", 7),
                       ("build", 1), ("build", 2), ("build", 4), ("build", 7), ("build", 9), ("build", 255), ("reserved0", 1)):
        t = _copy(base)
        setattr(t, field, bad)
        assert dga.tiling_check(t) == E_TILING, (field, bad)
    t = _copy(base); t.splitkFactor = 1025
    assert dga.tiling_check(t) == E_RANGE
    t = _copy(base); t.splitkFactor = 1024
    assert dga.tiling_check(t) == OK
    # what a field MAY hold beside the selector's pick
    for field, good in (("stages", 0), ("stages", 3), ("build", 0), ("wavesM", 0), ("dispatchPolicyTag", 0), ("dispatchPolicyTag", 1), ("dispatchPolicyTag", 2 | 16),
                        ("dispatchPolicyTag", 6), ("kernelSerial", 5), ("kernelSerial", 7), ("k1", 0)):
        t = _copy(base)
        setattr(t, field, good)
        if field == "wavesM":
            t.wavesN = 0
        assert dga.tiling_check(t) == OK, (field, good)
    # ping-pong, the quarter-tile tail and the one-launch Stream-K exist for the 256 x 256 tile only
    small = dga.tiling(1024, 2048, 7168)
    for field, v in (("dispatchPolicyTag", 1), ("kernelSerial", 5), ("kernelSerial", 7)):
        t = _copy(small)
        if (t.m1, t.n1) != (256, 256):
            setattr(t, field, v)
            assert dga.tiling_check(t) == E_TILING
    # build = 1 (DGA_BUILD_WSK_REGISTER) names the register build of the workgroup split-K and nothing else; `stages` is a stage count again
    t = _copy(base); t.build = 1; t.kernelSerial = 6
    assert dga.tiling_check(t) == OK
    t = _copy(base); t.stages = 1; t.kernelSerial = 6
    assert dga.tiling_check(t) == E_TILING
    # the bf16-exact policy's build names (include/dga_hip.h DGA_BUILD_BX_*): 4 .. 9; strict takes anything
    bx = dga.tiling(4096, 4096, 4096, policy="bf16_exact")
    for b, want in ((0, OK), (4, OK), (5, OK), (6, OK), (7, OK), (8, OK), (9, OK), (1, E_TILING), (2, E_TILING), (3, E_TILING), (10, E_TILING), (200, E_TILING)):
        t = _copy(bx); t.build = b
        assert dga.tiling_check(t) == want, b
    for st, want in ((0, OK), (2, OK), (3, OK), (1, E_TILING), (4, E_TILING), (7, E_TILING), (9, E_TILING), (200, E_TILING)):
        t = _copy(bx); t.stages = st
        assert dga.tiling_check(t) == want, st
    t = _copy(bx); t.build = 1; t.kernelSerial = 6
    assert dga.tiling_check(t) == OK
    # build 10 (DGA_BUILD_BX_DECODE): the one-launch split-K of the 64 x 128 tile -- a name of kernelSerial 6 on that tile and no other
    for ks, m1, n1, want in ((6, 64, 128, OK), (4, 64, 128, E_TILING), (0, 64, 128, E_TILING), (6, 128, 256, E_TILING), (6, 32, 128, E_TILING), (6, 64, 256, E_TILING)):
        t = _copy(bx); t.build, t.kernelSerial, t.m1, t.n1 = 10, ks, m1, n1
        assert dga.tiling_check(t) == want, (ks, m1, n1)
    t = _copy(base); t.build, t.kernelSerial = 10, 6          # (the fast path has no such build)
    assert dga.tiling_check(t) == E_TILING
    # ... its quarter-tile tail (kernelSerial 5) and its one-launch Stream-K (7) exist for the 128 x 256 tile only
    assert (bx.m1, bx.n1) == (128, 256)
    for ks, m1, n1, want in ((5, 128, 256, OK), (5, 64, 256, E_TILING), (5, 128, 128, E_TILING), (5, 32, 128, E_TILING), (7, 128, 256, OK),
                             (7, 128, 128, E_TILING), (7, 64, 256, E_TILING)):
        t = _copy(bx); t.kernelSerial, t.m1, t.n1 = ks, m1, n1
        assert dga.tiling_check(t) == want, (ks, m1, n1)
    t = _copy(base); t.dispatchPolicyTag = 3; t.m1 = 7; t.n1 = 9; t.stages = 77; t.wavesM = 5; t.build = 33
    assert dga.tiling_check(t) == OK


def test_the_bf16_exact_selector_names_its_quarter_tile_tail():
    """Rasters of 128 x 256 tiles between one and two rounds that leave a last round of at most half the CUs: the policy's own tiling
    names the launch pair (kernelSerial 5, blockDim = whole rounds + 4 quarter tiles per tail tile); from two rounds on a partial last
    round is cut by the one-launch Stream-K (kernelSerial 7, one workgroup per CU); whole rounds stay single launches."""
    for (m, n, k), tail in (((2304, 4096, 7168), 32), ((2560, 4096, 4096), 64), ((5120, 5120, 5120), 32)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        tiles = -(-m // 128) * -(-n // 256)
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor) == (128, 256, 5, 1) and tiles % 256 == tail, (m, n, k, t.as_dict())
        assert t.blockDim == tiles - tail + 4 * tail and dga.tiling_check(t) == OK
    for (m, n, k) in ((1024, 18432, 7168), (5119, 6997, 9901), (3511, 6151, 8191)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.blockDim) == (128, 256, 7, 1, 256), (m, n, k, t.as_dict())
        assert dga.tiling_check(t) == OK
    for (m, n, k) in ((4096, 4096, 4096), (8192, 8192, 8192), (4096, 2048, 7168), (1279, 5120, 7680)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert t.kernelSerial not in (5, 7), (m, n, k, t.as_dict())


def test_fuzzed_structs_are_either_in_the_menu_or_refused():
    """Random field values: the check never crashes, is deterministic, and what it accepts on the fast path is a tile, a wave grid
    and a stage count the menu holds."""
    rng = random.Random(5)
    base = dga.tiling(2048, 4096, 7168)
    accepted = 0
    for _ in range(20000):
        t = _copy(base)
        t.kernelSerial = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 255])
        t.dispatchPolicyTag = rng.choice([0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 16, 18, 20, 22, 23, 24, 31, 64, 255])
        t.m1 = rng.choice([0, 8, 16, 32, 48, 64, 96, 128, 256, 512, 65535])
        t.n1 = rng.choice([0, 64, 128, 192, 256, 512])
        t.k1 = rng.choice([0, 128, 128, 128, 64, 256])
        t.wavesM = rng.choice([0, 0, 1, 2, 3, 4, 8]); t.wavesN = rng.choice([0, 0, 1, 2, 4, 8])
        t.stages = rng.choice([0, 0, 1, 2, 2, 3, 3, 4, 7, 9, 255])
        t.build = rng.choice([0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 255])
        t.reserved0 = rng.choice([0, 0, 0, 0, 0, 0, 0, 1, 255])
        t.splitkFactor = rng.choice([0, 1, 2, 4, 8, 56, 1024, 1025, 65535])
        rc = dga.tiling_check(t)
        assert rc in (OK, E_TILING, E_RANGE) and rc == dga.tiling_check(t)
        if rc == OK:
            accepted += 1
            tag = t.dispatchPolicyTag & 7
            assert t.kernelSerial in (0, 1, 2, 4, 5, 6, 7) and t.k1 in (0, 128) and t.splitkFactor <= 1024 and not (t.dispatchPolicyTag & ~23)
            assert t.reserved0 == 0
            if tag != 3:
                assert t.stages in (0, 2, 3) and (t.build != 1 or t.kernelSerial == 6)
            if tag not in (3, 7):
                assert (t.m1, t.n1) in FAST_TILES and t.build in (0, 1)
            if tag == 7:
                assert t.build in (0, 1, 4, 5, 6, 7, 8, 9, 10) and (t.build != 10 or (t.kernelSerial, t.m1, t.n1) == (6, 64, 128))
    assert 0 < accepted < 20000


REFERENCE_ROWS = ("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim\n"
                  "512,512,512,128,256,256,0,0,0,0,24\n"          # the reference's own fixture rows (csv_test.cpp:33-35): k1 = 256
                  "1024,1024,1024,256,256,256,1,1,0,0,24\n"
                  "2048,2048,2048,128,256,512,3,1,1,0,24\n"       # kernel type 3 (PaddingStreamK): no build of that name here
                  "4096,4096,4096,256,128,1024,4,0,0,0,24\n"
                  "300,520,1024,128,128,256,0,0,0,0,15\n")


def test_tilings_from_a_reference_format_cache_pass_the_check(tmp_path):
    """With a reference-format CSV open (cache.cpp:22-101), what dga_tiling and dga_tiling_bf16_exact hand back for a cached shape is
    a tiling the launchers accept: k1 and kernel types the menu lacks are normalised on the way out of the cache, not refused at the
    launch (they were, for every cached shape)."""
    path = tmp_path / "ref.csv"
    path.write_text(REFERENCE_ROWS)
    try:
        dga.tiling_cache_open(str(path))
        assert dga.tiling_cache_size() == 5
        for m, n, k in ((512, 512, 512), (1024, 1024, 1024), (2048, 2048, 2048), (4096, 4096, 4096), (300, 520, 1024)):
            for pol in (None, "bf16_exact", "strict"):
                t = dga.tiling(m, n, k, policy=pol)
                assert dga.tiling_check(t) == OK, (m, n, k, pol, t.as_dict())
                assert t.k1 in (0, 128) and t.kernelSerial in (0, 1, 2, 4, 5, 6, 7)
        t = dga.tiling(512, 512, 512)
        assert (t.m1, t.n1, t.blockDim) == (128, 256, 24)      # the tile is still the file's
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_null_is_an_error_not_a_crash():
    assert _lib.lib().dga_tiling_check(None) == -1
