"""kernelSerial 7 under the bf16-exact policy: Stream-K in one launch on the persistent 128 x 256 kernel
(csrc/gemm_fp8_bf16x_streamk_kernel.hpp) -- whole rounds as the persistent kernel runs them, the last partial round cut along K:
s equal k ranges per tile when it holds at most half a round of tiles, mains [0, k*) + tails [k*, KB) above that; fp32 partial tiles
through the workspace, added in k order by one piece of each tile.  The reference's kernel type 4
(/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:94-98, selection rule
op_host/op_tiling/select_kernel.cpp:303-331).  The policy's bar against the CPU oracle, every element against the persistent kernel
(same arithmetic per k block, partial sums regrouped at the cuts), determinism, graph replay, ragged edges, the fall-backs."""
import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev

pytestmark = pytest.mark.gpu


def _tiling(dga, m, n, k, streamk=True):
    t = dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.splitkFactor, t.dispatchPolicyTag = 128, 256, 0, 0, 3, 1, 7
    t.kernelSerial, t.build = (7, 0) if streamk else (0, 7)
    assert dga.tiling_check(t) == 0
    return t


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


# (tiles_m, tiles_n, K): on 256 CUs -- 1 tile (8 ranges), 32 tiles (8), 64 (4), 100 (2), 128 (2), 129 (mains + 2 tails... t = 2),
# 200 (t = 4), 255 (t = 255: one spare workgroup walks 255 tails), 1.5 rounds (form 1 behind a whole round), 2.73 rounds (form 2
# behind two), exact rounds (nothing to cut: the fall-back), ragged edges, K % 128 != 0, short K
SHAPES = [(1, 1, 2048, 0, 0), (4, 8, 1024, 0, 0), (8, 8, 1024, 0, 0), (10, 10, 768, 0, 0), (8, 16, 7168, 0, 0), (3, 43, 640, 0, 0),
          (10, 20, 1280, 0, 0), (15, 17, 512, 0, 0), (24, 16, 512, 0, 0), (28, 25, 640, 0, 0), (16, 16, 512, 0, 0),
          (10, 20, 1280 + 16, -7, -100), (5, 21, 256 + 64, -127, -255), (9, 9, 512, -1, -1)]


@pytest.mark.parametrize("tm,tn,k,dm,dn", SHAPES)
def test_parity_determinism_and_the_persistent_kernel(dga, oracle, tm, tn, k, dm, dn):
    m, n = 128 * tm + dm, 256 * tn + dn
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=tm * 100 + tn + k)
    t = _tiling(dga, m, n, k)
    assert dga.workspace_bytes(t) >= _cus() * (128 * 256 * 4 + 8)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
    got = _bits(out)
    assert not np.isnan(out.float().cpu().numpy()).any(), "an output element was never written"
    rows = sorted(set(list(range(0, min(m, 40))) + list(range(m // 2, min(m, m // 2 + 16))) + list(range(max(0, m - 40), m))))
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    _assert_bar(oracle, got[rows], want, a[rows], sfa[rows], b, sfb)
    ref = torch.empty_like(out)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=_tiling(dga, m, n, k, streamk=False), sync=True)
    d = oracle.bf16_ulp_diff(got, _bits(ref))
    # (regrouped fp32 sums move an output only where its sum cancels: a handful of elements by one bf16 ulp)
    assert float((d > 0).mean()) < 2e-3 and float((d > 1).mean()) < 2e-4, (int(d.max(initial=0)), float((d > 0).mean()))
    out2 = torch.empty_like(out)
    for _ in range(3):      # deterministic: the order of the additions is fixed, whatever the timing
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out2, tiling_=t, sync=True)
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_exact_rounds_and_missing_workspace_fall_back_to_the_tile_kernels(dga, oracle):
    m, n, k = 128 * 16, 256 * 16, 512         # 256 tiles: nothing to cut
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=5)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    outs = []
    for streamk in (True, False):
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o, tiling_=_tiling(dga, m, n, k, streamk), sync=True)
        outs.append(o)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    # the C ABI without a workspace: the same tiling runs the tile kernel (same bits as the persistent build on a shape that has a cut)
    from deepgemm_ascend_amd import _lib
    import ctypes
    m, n = 128 * 10, 256 * 10
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=6)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    t = _tiling(dga, m, n, k)
    o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(ta.data_ptr(), tsa.data_ptr(), tb.data_ptr(), tsb.data_ptr(), o.data_ptr(), m, n, k,
                                             ctypes.byref(t), None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    ref = torch.empty_like(o)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=_tiling(dga, m, n, k, streamk=False), sync=True)
    assert torch.equal(o.view(torch.int16), ref.view(torch.int16))


def test_graph_replay(dga, oracle):
    """The reader of a flag puts it back to 0, so a captured launch (replayed with its epoch) needs no memset node: three replays give
    the direct launch's bytes."""
    m, n, k = 128 * 10, 256 * 20, 1280
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=9)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    t = _tiling(dga, m, n, k)
    direct = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), direct, tiling_=t, sync=True)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)   # (workspace allocation outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)
    for _ in range(3):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), direct.view(torch.int16))


def test_the_selector_names_it(dga):
    """dga_tiling_bf16_exact names kernelSerial 7 where it measured ahead: rasters of 128 x 256 tiles of at least two rounds with a
    partial last one and K >= 6144 (select_kernel.cpp:303-331: more blocks than cores with a remainder, deep K).  Below two rounds the launch pair,
    the persistent kernel or smaller tiles stay (profiles/r06_bx_streamk.txt)."""
    for m, n, k in ((3511, 6151, 8191), (1024, 18432, 7168), (5119, 6997, 9901)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor) == (128, 256, 7, 1), (m, n, k, t.as_dict())
        assert dga.tiling_check(t) == 0 and dga.workspace_bytes(t) >= _cus() * (128 * 256 * 4 + 8)
    # whole rounds, fewer than two rounds, short K (the adding pass is a fixed cost: profiles/r06_bx_regret.txt)
    for m, n, k in ((4096, 4096, 4096), (8192, 8192, 8192), (1024, 4096, 7168), (1279, 5003, 7681), (2304, 4096, 7168), (5120, 5120, 5120),
                    (4096, 7168, 2048)):
        assert dga.tiling(m, n, k, policy="bf16_exact").kernelSerial != 7


def test_fuzzed_rasters(dga, oracle):
    """Seeded fuzz over dense shapes whose rasters fall into every form of the cut (none, s ranges, mains + tails with 1..many tails per
    spare) with ragged edges and K tails: the policy's bar on sampled rows, every element against the persistent kernel, determinism."""
    rng = np.random.default_rng(77)
    for case in range(12):
        tm, tn = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        k = 128 * int(rng.integers(2, 14)) + int(rng.choice([0, 0, 16, 80]))
        m = 128 * tm - int(rng.integers(0, 100)) if tm > 1 else 128 - int(rng.integers(0, 100))
        n = 256 * tn - int(rng.integers(0, 200)) if tn > 1 else 256 - int(rng.integers(0, 200))
        a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=500 + case)
        ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
        t = _tiling(dga, m, n, k)
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
        what = f"case {case}: {m} x {n} x {k} ({tm * tn} tiles)"
        assert not bool(torch.isnan(out.float()).any()), what + ": an output element was never written"
        ref = torch.empty_like(out)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=_tiling(dga, m, n, k, streamk=False), sync=True)
        d = oracle.bf16_ulp_diff(_bits(out), _bits(ref))
        assert float((d > 0).mean()) < 3e-3 and float((d > 1).mean()) < 3e-4, (what, int(d.max(initial=0)), float((d > 0).mean()))
        rows = sorted(set(list(range(0, min(m, 24))) + list(range(max(0, m - 24), m))))
        want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
        _assert_bar(oracle, _bits(out)[rows], want, a[rows], sfa[rows], b, sfb)
        out2 = torch.empty_like(out)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out2, tiling_=t, sync=True)
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), what + ": two launches differ"
