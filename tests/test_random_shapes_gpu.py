"""Round 6's new kernels on random shapes and arbitrary e4m3 bytes: the one-launch decode split-K (kernelSerial 6, build 10) against the
one-pass tile kernel -- ULP fractions (the same slices summed in another grouping), determinism, no unwritten element -- and the masked
grouped kernel (build 9) against the one-tile build, bit for bit.  Seeded; the shapes cover ragged N, K % 128 != 0, 1..512 rows, every
split count the launcher takes, allocations of 64..256 rows per expert."""
import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _bits, _dev

pytestmark = pytest.mark.gpu


def _bytes(rng, shape):
    x = rng.integers(0, 256, size=shape, dtype=np.uint8)
    x[(x & 0x7F) == 0x7F] = 0x3C          # (no NaN codes: NaN placement has its own tests)
    return x


def test_decode_build_on_random_shapes(dga, oracle):
    from deepgemm_ascend_amd import _lib
    rng = np.random.default_rng(2024)
    done = 0
    while done < 70:
        m = int(rng.integers(1, 513)); n = int(rng.integers(1, 64)) * 128 - int(rng.integers(0, 128)); k = int(rng.integers(4, 150)) * 128 - 16 * int(rng.integers(0, 8))
        kb, tiles = -(-k // 128), -(-m // 64) * -(-n // 128)
        if tiles > 256:
            continue
        done += 1
        s = max(1, min(int(rng.integers(1, 9)), 256 // tiles, kb // 4))
        a, b = _bytes(rng, (m, k)), _bytes(rng, (n, k))
        sfa = rng.uniform(0.25, 2.0, size=(m, kb)).astype(np.float32)
        sfb = rng.uniform(0.25, 2.0, size=(-(-n // 128), kb)).astype(np.float32)
        ta, tsa, tb, tsb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.splitkFactor, t.dispatchPolicyTag, t.swizzleOffset = 64, 128, 0, 0, 3, s, 7, 1
        t.kernelSerial, t.build = 6, _lib.BUILD_BX_DECODE
        outs = []
        for _ in range(2):
            o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
            dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o, tiling_=t, sync=True)
            outs.append(o)
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (m, n, k, s)
        assert not bool(torch.isnan(outs[0].float()).any()), (m, n, k, s)
        r = dga.tiling(m, n, k, policy="bf16_exact")
        r.m1, r.n1, r.wavesM, r.wavesN, r.stages, r.splitkFactor, r.dispatchPolicyTag, r.kernelSerial, r.build = 64, 128, 0, 0, 3, 1, 7, 0, 0
        ref = torch.empty_like(outs[0])
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=r, sync=True)
        d = oracle.bf16_ulp_diff(_bits(outs[0]), _bits(ref))
        assert float((d > 0).mean()) < 6e-3 and float((d > 1).mean()) < 1e-3, (m, n, k, s, float((d > 0).mean()), int(d.max(initial=0)))


def test_grouped_kernel_on_random_layouts(dga):
    rng = np.random.default_rng(7)
    for _ in range(24):
        g = int(rng.integers(2, 40)); mmax = int(rng.choice([64, 96, 128, 160, 200, 256]))
        n = int(rng.integers(1, 10)) * 256 - int(rng.integers(0, 2)) * int(rng.integers(0, 255)); k = int(rng.integers(2, 40)) * 128 - 16 * int(rng.integers(0, 8))
        kb = -(-k // 128)
        a, b = _bytes(rng, (g, mmax, k)), _bytes(rng, (g, n, k))
        sfa = rng.uniform(0.25, 2.0, size=(g, mmax, kb)).astype(np.float32)
        sfb = rng.uniform(0.25, 2.0, size=(g, -(-n // 128), kb)).astype(np.float32)
        masked = rng.integers(0, mmax + 1, size=(g,)).astype(np.int32)
        ta, tsa, tb, tsb, tm = _dev(a), _dev(sfa), _dev(b), _dev(sfb), _dev(masked)
        outs = []
        for build in (9, 8):
            t = dga.tiling(mmax, n, k, groups=g, expected_m=mmax, policy="bf16_exact")
            t.m1, t.n1, t.build, t.kernelSerial, t.splitkFactor = 128, 256, build, 0, 1
            o = torch.full((g, mmax, n), -7.0, dtype=torch.bfloat16, device="cuda")
            dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ta, tsa), (tb, tsb), o, tm, mmax, policy="bf16_exact", tiling_=t, sync=True)
            outs.append(o)
        assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), (g, mmax, n, k)
