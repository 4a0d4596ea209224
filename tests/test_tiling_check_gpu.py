"""GPU half of tests/test_tiling_check.py: a fuzzed dga_tiling_t handed to the operator either runs a build of the menu and gives
the right answer, or is refused with the status dga_tiling_check names -- before any launch (the output buffer keeps its bytes)."""
import ctypes
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("m,n,k", [(300, 520, 1024), (48, 1030, 2048)])
def test_fuzzed_tilings_run_correctly_or_are_refused(dga, oracle, m, n, k):
    from deepgemm_ascend_amd import _lib
    # power-of-two block scales: a tiling may carry DGA_POLICY_UE8M0_SCALES (16), the caller's promise that they are
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + k, ue8m0=True)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    ta, tsa, tb, tsb = (torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb))
    rng = random.Random(m)
    base = dga.tiling(m, n, k)
    ran = refused = 0
    for _ in range(240):
        t = _lib.Tiling()
        ctypes.memmove(ctypes.byref(t), ctypes.byref(base), ctypes.sizeof(_lib.Tiling))
        t.kernelSerial = rng.choice([0, 0, 1, 2, 4, 5, 6, 7, 3, 8])
        t.dispatchPolicyTag = rng.choice([0, 1, 2, 4, 5, 6, 7, 3, 16, 18, 20, 8, 32])
        t.m1 = rng.choice([16, 32, 64, 128, 256, 48, 0])
        t.n1 = rng.choice([128, 256, 64])
        t.wavesM, t.wavesN = rng.choice([(0, 0), (0, 0), (1, 4), (2, 2), (2, 4), (4, 2), (4, 1), (3, 3)])
        t.stages = rng.choice([0, 2, 3, 2, 3, 2, 3, 2, 3, 1, 4])
        t.build = rng.choice([0] * 12 + [1, 4, 5, 6, 7, 8, 9, 2])
        t.splitkFactor = rng.choice([1, 1, 1, 2, 4, 8])
        rc = dga.tiling_check(t)
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        if rc != 0:
            with pytest.raises(dga.DGAError) as e:
                dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
            assert e.value.status == rc
            torch.cuda.synchronize()
            assert bool(torch.isnan(out).all()), "a refused tiling wrote to the output"
            refused += 1
            continue
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
        got = _bits(out)
        what = (f"kernelSerial {t.kernelSerial} policy {t.dispatchPolicyTag} tile {t.m1}x{t.n1} waves {t.wavesM}x{t.wavesN} "
                f"stages {t.stages} build {t.build} splitk {t.splitkFactor}")
        try:
            if (t.dispatchPolicyTag & 7) == 3:
                assert np.array_equal(got, want)
            else:
                oracle.assert_parity(got, want, a, sfa, b, sfb)
        except AssertionError as e:
            raise AssertionError(f"{what}: {e}") from None
        ran += 1
    assert ran >= 20 and refused >= 20, (ran, refused)


def test_a_gemm_on_a_cached_reference_row_runs(dga, oracle, tmp_path):
    """A reference-format cache file (k1 = 256, a kernel type this menu lacks) is open: default calls (tiling == NULL in the C ABI) and
    calls that pass dga.tiling(...) run and are right, under both arithmetic policies."""
    from test_tiling_check import REFERENCE_ROWS
    path = tmp_path / "ref.csv"
    path.write_text(REFERENCE_ROWS)
    m, n, k = 300, 520, 1024
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=3)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    ta, tsa, tb, tsb = (torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb))
    try:
        dga.tiling_cache_open(str(path))
        dga.api._PLANS.clear()
        for pol in ("fast", "bf16_exact"):
            for explicit in (False, True):
                out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
                t = dga.tiling(m, n, k, policy=pol) if explicit else None
                dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, policy=pol, tiling_=t, sync=True)
                oracle.assert_parity(_bits(out), want, a, sfa, b, sfb)
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()
        dga.api._PLANS.clear()
