"""kernelSerial 6 with build DGA_BUILD_BX_DECODE under the bf16-exact policy: the one-launch split-K of the 64 x 128 tile
(csrc/gemm_fp8_bf16x_dsk_kernel.hpp) -- two k groups of four waves per workgroup on their own LDS rings, splitkFactor workgroups per
tile whose fp32 partial tiles meet in the workspace and are added in k order by the workgroup that holds the first slices.  The
reference's split-K kernel types (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36)
and the fused reduce of its Stream-K kernel (op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107).  The policy's bar against the
CPU oracle, every element against the two-launch split-K, determinism, graph replay without a memset node, ragged edges, fall-backs."""
import ctypes

import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev

pytestmark = pytest.mark.gpu


def _tiling(dga, m, n, k, s, decode=True):
    from deepgemm_ascend_amd import _lib
    t = dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.splitkFactor, t.dispatchPolicyTag, t.swizzleOffset = 64, 128, 0, 0, 3, s, 7, 1
    t.kernelSerial, t.build = (6, _lib.BUILD_BX_DECODE) if decode else ((4 if s > 1 else 0), 0)
    assert dga.tiling_check(t) == 0
    return t


# (m, n, k, splits): one and two tile rows, at most 32 rows (the two-m-tile loop), ragged N, K % 128 != 0, two k blocks per slice,
# uneven slices (the shorter k group idles through a barrier), the adding workgroup's longer slices, 256 workgroups
SHAPES = [(64, 4096, 7168, 7), (128, 4096, 7168, 4), (33, 2112, 1024, 2), (17, 640, 2048, 4), (1, 384, 1024, 1), (64, 24576, 1536, 1),
          (100, 1000, 4096 + 48, 5), (64, 128, 8192, 8), (128, 2112, 7168, 7), (48, 4096, 1024, 2), (65, 8192, 3328, 2), (96, 4096, 4096, 4),
          (64, 3968, 512, 1), (128, 16384, 7168, 1), (63, 100, 640, 1)]


@pytest.mark.parametrize("m,n,k,s", SHAPES)
def test_parity_determinism_and_the_two_launch_split(dga, oracle, m, n, k, s):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + n + k + s)
    t = _tiling(dga, m, n, k, s)
    tiles = -(-m // 64) * -(-n // 128)
    if s > 1:
        assert dga.workspace_bytes(t) >= tiles * (s - 1) * (64 * 128 * 4 + 8)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
    got = _bits(out)
    assert not np.isnan(out.float().cpu().numpy()).any(), "an output element was never written"
    rows = sorted(set(list(range(0, min(m, 24))) + list(range(max(0, m - 24), m))))
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    _assert_bar(oracle, got[rows], want, a[rows], sfa[rows], b, sfb)
    ref = torch.empty_like(out)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=_tiling(dga, m, n, k, 2 * s, decode=False), sync=True)
    d = oracle.bf16_ulp_diff(got, _bits(ref))
    # (the same slices summed in another grouping: an output moves only where its sum cancels, by one bf16 ulp)
    assert float((d > 0).mean()) < 2e-3 and float((d > 1).mean()) < 2e-4, (int(d.max(initial=0)), float((d > 0).mean()))
    out2 = torch.empty_like(out)
    for _ in range(3):      # deterministic: the order of the additions is fixed, whatever the timing
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out2, tiling_=t, sync=True)
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16))


def test_one_split_is_the_two_launch_split_by_two_bit_for_bit(dga, oracle):
    """splitkFactor 1: the workgroup's two k groups are the two slabs of the two-launch split-K by 2 (same slices where K is an even
    number of k blocks, same arithmetic inside a slice, one addition): the same bits."""
    for m, n, k in ((64, 1024, 2048), (128, 2112, 1024), (40, 640, 512)):
        a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=k + m)
        ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
        o1, o2 = (torch.empty((m, n), dtype=torch.bfloat16, device="cuda") for _ in range(2))
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o1, tiling_=_tiling(dga, m, n, k, 1), sync=True)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o2, tiling_=_tiling(dga, m, n, k, 2, decode=False), sync=True)
        assert torch.equal(o1.view(torch.int16), o2.view(torch.int16)), (m, n, k)


def test_what_it_does_not_take_runs_the_two_launch_split(dga, oracle):
    """More tiles than CUs, fewer than four k blocks, no workspace: the same tiling runs the tile kernels (the C ABI's documented fall-back)."""
    from deepgemm_ascend_amd import _lib
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for m, n, k, s in ((64, 128 * (cus + 3), 1024, 1), (64, 2048, 384, 1), (128, 4096, 2048, 4)):
        a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=n)
        ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
        t = _tiling(dga, m, n, k, s)
        o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        if s > 1:      # no workspace at all: neither the partial tiles nor the slabs -- one pass of the tile kernel
            rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(ta.data_ptr(), tsa.data_ptr(), tb.data_ptr(), tsb.data_ptr(), o.data_ptr(), m, n, k,
                                                     ctypes.byref(t), None, 0, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            ref_t = _tiling(dga, m, n, k, 1, decode=False)
        else:
            dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o, tiling_=t, sync=True)
            ref_t = _tiling(dga, m, n, k, 1, decode=False)
        ref = torch.empty_like(o)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=ref_t, sync=True)
        assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), (m, n, k, s)


def test_graph_replay_needs_no_memset(dga, oracle):
    """The reader of a flag puts it back to 0, so a captured launch -- replayed with the same epoch -- finds its flags as a direct
    launch does: five replays give the direct launch's bytes, and so does a direct launch in between."""
    m, n, k, s = 128, 4096, 7168, 4
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=9)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    t = _tiling(dga, m, n, k, s)
    direct = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), direct, tiling_=t, sync=True)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)   # (workspace allocation outside the capture)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)   # two launches of one graph share the workspace in stream order
    for i in range(5):
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int16), direct.view(torch.int16))
        if i == 2:
            out.zero_()
            dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)
            assert torch.equal(out.view(torch.int16), direct.view(torch.int16))


def test_the_selector_names_it_and_a_default_call_runs_it(dga, oracle):
    """dga_tiling_bf16_exact names the build where it measured ahead of every other candidate under the decode protocol
    (profiles/r06_decode_cold_table.txt): 33..512 rows (17..32 on matrices of more than 8192 rows or with K of at least 12288), 24 tiles to one per CU, at most 30 k
    blocks per k group (16 above 128 rows), no thin grid with long k groups;
    the neighbours keep their picks."""
    from deepgemm_ascend_amd import _lib
    for m, n, k, s in ((128, 4096, 7168, 4), (128, 2112, 7168, 7), (96, 7168, 2048, 2), (128, 4096, 4096, 4), (64, 4096, 4096, 6), (64, 4096, 7168, 6),
                       (256, 4096, 7168, 2), (40, 7168, 4096, 4), (64, 24576, 1536, 1), (192, 2112, 7168, 5), (64, 7168, 16384, 4), (32, 24576, 1536, 1),
                       (24, 12288, 5120, 2), (24, 4096, 18432, 6), (48, 3072, 18432, 8), (512, 2048, 7168, 2), (320, 2112, 7168, 3)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert (t.m1, t.n1, t.kernelSerial, t.build, t.splitkFactor) == (64, 128, 6, _lib.BUILD_BX_DECODE, s), (m, n, k, t.as_dict())
        assert dga.tiling_check(t) == 0
    for m, n, k in ((64, 2112, 7168), (128, 7168, 18432), (128, 8192, 1024), (256, 5120, 5120), (64, 4096, 2048), (32, 4096, 7168), (300, 4096, 7168),
                    (128, 4096, 7160 + 4), (64, 32768 + 128, 1536), (16, 24576, 1536), (64, 18432, 7168)):
        assert dga.tiling(m, n, k, policy="bf16_exact").build != _lib.BUILD_BX_DECODE, (m, n, k)
    m, n, k = 128, 4096, 7168
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=3)
    ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
    o1, o2 = (torch.empty((m, n), dtype=torch.bfloat16, device="cuda") for _ in range(2))
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o1, sync=True)                                   # the default call
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o2, tiling_=_tiling(dga, m, n, k, 4), sync=True)
    assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))


def test_fuzz_against_the_two_launch_split(dga, oracle):
    rng = np.random.default_rng(77)
    for _ in range(14):
        m = int(rng.integers(1, 200)); n = int(rng.integers(1, 40)) * 128 - int(rng.integers(0, 128)); k = int(rng.integers(4, 60)) * 128 - 16 * int(rng.integers(0, 8))
        kb = -(-k // 128)
        tiles = -(-m // 64) * -(-n // 128)
        s = int(rng.integers(1, 9))
        s = max(1, min(s, 256 // tiles, kb // 4))
        a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=int(rng.integers(1 << 30)))
        ta, tsa, tb, tsb = (_dev(x) for x in (a, sfa, b, sfb))
        o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), o, tiling_=_tiling(dga, m, n, k, s), sync=True)
        ref = torch.empty_like(o)
        dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), ref, tiling_=_tiling(dga, m, n, k, 1, decode=False), sync=True)
        d = oracle.bf16_ulp_diff(_bits(o), _bits(ref))
        assert float((d > 0).mean()) < 4e-3 and float((d > 1).mean()) < 5e-4, (m, n, k, s, int(d.max(initial=0)), float((d > 0).mean()))
