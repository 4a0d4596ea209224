"""kernelSerial 7: Stream-K in one launch (csrc/gemm_fp8_streamk_kernel.hpp) -- whole rounds of 256 x 256 tiles as in the
persistent kernel, the last partial round split along K with fp32 partial tiles through the workspace and a parallel reduction
in k order.  The reference's kernel type 4 (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/
padding_streamk_matmul_kernel.h:94-98, selection rule op_host/op_tiling/select_kernel.cpp:303-331).  Parity against the CPU oracle
under the fast policy's bar, determinism, the hardware-scale form, graph replay, and the fall-back for shapes it does not take."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _tiling(dga, m, n, k, flag=0):
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 256, 256, 4, 2, 2, 2 | flag, 7, 1
    assert dga.tiling_check(t) == 0
    return t


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


def _shapes():
    """(tiles_m, tiles_n, k): parts per tail tile 16, 8, 4, 2 and 1, whole rounds + a tail, exact rounds, a single tile."""
    return [(1, 1, 4096), (2, 8, 2048), (4, 8, 1024), (8, 8, 1024), (10, 16, 512), (16, 16, 512), (17, 16, 512), (16, 18, 1024),
            (9, 32, 512), (3, 1, 256)]


@pytest.mark.parametrize("tm,tn,k", _shapes())
def test_parity_on_sampled_rows(dga, oracle, tm, tn, k):
    m, n = 256 * tm, 256 * tn
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=tm * 100 + tn)
    t = _tiling(dga, m, n, k)
    assert dga.workspace_bytes(t) >= _cus() * 256 * 256 * 4
    ta, tsa, tb, tsb = (torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb))
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
    got = _bits(out)
    assert not np.isnan(out.float().cpu().numpy()).any(), "an output element was never written"
    # rows of the first tile, of a middle tile row and of the last (tail) tile row
    rows = sorted(set(list(range(0, 48)) + list(range(m // 2, m // 2 + 16)) + list(range(m - 48, m))))
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    oracle.assert_parity(got[rows], want, a[rows], sfa[rows], b, sfb)
    # every element against the one-tile kernel: the same arithmetic per k block, partial sums regrouped at the part boundaries
    t1 = dga.tiling(m, n, k)
    t1.m1, t1.n1, t1.wavesM, t1.wavesN, t1.stages, t1.dispatchPolicyTag, t1.kernelSerial, t1.splitkFactor = 256, 256, 4, 2, 2, 2, 0, 1
    ref = torch.empty_like(out)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=t1, sync=True)
    d = oracle.bf16_ulp_diff(got, _bits(ref))
    # (regrouped fp32 sums move an output only where the sum cancels: a handful of elements, by a few bf16 ulp)
    assert float((d > 0).mean()) < 2e-3 and float((d > 1).mean()) < 2e-4, (int(d.max(initial=0)), float((d > 0).mean()))
    # deterministic: a second launch gives the same bytes (the reduction order is fixed, whatever the timing)
    out2 = torch.empty_like(out)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out2, tiling_=t, sync=True)
    assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_hardware_scale_form_and_graph_replay(dga, oracle):
    m, n, k = 256 * 17, 256 * 16, 1024      # one whole round + a tail split 16 ways on a 256-CU part
    rng = np.random.default_rng(3)
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=5, ue8m0=True)
    ta, tsa, tb, tsb = (torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb))
    t = _tiling(dga, m, n, k, flag=16)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
    rows = list(range(m - 300, m))
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    oracle.assert_parity(_bits(out)[rows], want, a[rows], sfa[rows], b, sfb)
    # captured into a graph and replayed on CHANGED operands: the flags of the previous replay must not be taken for this one's
    t0 = _tiling(dga, m, n, k)
    g_out = torch.empty_like(out)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), g_out, tiling_=t0)      # (workspace allocated outside the capture)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), g_out, tiling_=t0)
    for seed in (11, 12, 13):
        a2, sfa2, _, _ = oracle.make_inputs(m, n, k, seed=seed)
        ta.copy_(torch.from_numpy(a2).cuda()); tsa.copy_(torch.from_numpy(sfa2).cuda())
        graph.replay()
        torch.cuda.synchronize()
        want = oracle.gemm_fp8_fp8_bf16_nt(a2[rows], sfa2[rows], b, sfb, threads=8)
        oracle.assert_parity(_bits(g_out)[rows], want, a2[rows], sfa2[rows], b, sfb)


def test_shapes_it_does_not_take_run_the_tile_kernel(dga, oracle):
    """Ragged M / N / K: the launcher answers DGA_E_TILING internally and the tiling's tile kernel runs -- same call, right answer."""
    m, n, k = 300, 520, 1040
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=8)
    t = _tiling(dga, m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt(tuple(torch.from_numpy(x).cuda() for x in (a, sfa)), tuple(torch.from_numpy(x).cuda() for x in (b, sfb)), out,
                             tiling_=t, sync=True)
    oracle.assert_parity(_bits(out), oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)


def test_the_selector_names_it_for_very_deep_k(dga, oracle):
    """dga_tiling (fast policy) names the one-launch Stream-K for rasters of at most 64 tiles of 256 x 256 with K >= 32768
    (profiles/r05_streamk_class_sweep.txt: -2..-21 % there; select_kernel.cpp:303-331 is the reference's rule for its kernel type 4),
    the learned predictor leaves the pick alone, and a default fast call on such a shape runs it and is right."""
    for m, n, k in ((256, 4096, 32768), (512, 7168, 32768), (1024, 4096, 32768)):
        t = dga.tiling(m, n, k, policy="fast")
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.blockDim) == (256, 256, 7, 1, _cus()), (m, n, k, t.as_dict())
        assert dga.tiling_check(t) == 0
    for m, n, k in ((256, 4096, 16384), (1024, 7168, 32768), (4096, 4096, 4096), (300, 4096, 32768)):   # shallower K, more tiles, whole rounds, ragged M
        assert dga.tiling(m, n, k, policy="fast").kernelSerial != 7
    m, n, k = 256, 1024, 32768
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=11)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt(tuple(torch.from_numpy(x).cuda() for x in (a, sfa)), tuple(torch.from_numpy(x).cuda() for x in (b, sfb)), out,
                             policy="fast", sync=True)
    rows = list(range(0, 32)) + list(range(m - 32, m))
    oracle.assert_parity(_bits(out)[rows], oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8), a[rows], sfa[rows], b, sfb)
