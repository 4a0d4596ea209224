"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star "within 2 ULP bf16"), as oracle.assert_parity states it:
|got - want| <= 2 ulp_bf16(want) + eps * S, S = sum |scaled products|, eps = 2^-15; the eps term is the
fp8 MFMA datapath's internal alignment error (DESIGN.md "Numerics"), needed by < 2e-3 of the elements
(outputs that cancel to near zero).  NaN positions must coincide."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MAX_ULP = 2


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t


def _run(dga, a, sfa, b, sfb, tiling=None):
    m, n = a.shape[0], b.shape[0]
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, tiling_=tiling, sync=True)
    return out.view(torch.int16).cpu().numpy().view(np.uint16)


def _check(oracle, got, a, sfa, b, sfb, threads=8, **kw):
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=threads)
    return oracle.assert_parity(got, want, a, sfa, b, sfb, **kw)


@pytest.mark.parametrize("m,n,k", [
    (128, 128, 128),      # BASELINE config 1 shape
    (64, 256, 384),
    (256, 256, 512),
    (300, 200, 256),      # M, N tails
    (1, 128, 128),        # M = 1 (the reference's special case, generate_code.hpp:233-236)
    (7, 136, 1024),       # N not a multiple of 8: scalar stores
    (129, 257, 144),      # K tail chunk (K % 128 = 16), every axis ragged
    (512, 384, 7168),     # DeepSeek K
    (33, 4096, 512),
])
def test_dense_parity_scaled(dga, oracle, m, n, k):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m * 31 + n * 7 + k)
    got = _run(dga, a, sfa, b, sfb)
    _check(oracle, got, a, sfa, b, sfb)


def test_config1_unit_scales_matches_reference_golden_formula(dga, oracle):
    """BASELINE config 1: 128^3, unit scales.  K = 128 is one scale block, so the oracle is exactly the
    reference golden np.matmul(f32, f32) (test.py:37) rounded to bf16."""
    a, sfa, b, sfb = oracle.make_inputs(128, 128, 128, seed=0, unit_scales=True)
    got = _run(dga, a, sfa, b, sfb)
    tab = oracle.np_e4m3fn_table()
    golden = np.matmul(tab[a].astype(np.float32), tab[b].astype(np.float32).T).astype(np.float32)
    oracle.assert_parity(got, oracle.f32_to_bf16_bits(golden), a, sfa, b, sfb)
    _check(oracle, got, a, sfa, b, sfb)


@pytest.mark.parametrize("bm,bn", [(256, 256), (128, 256), (256, 128), (128, 128), (64, 256), (64, 128),
                                   (32, 256), (32, 128), (16, 256), (16, 128)])
def test_every_kernel_variant(dga, oracle, bm, bn):
    """Force each compiled tile variant on a ragged problem (tails in M, N and K)."""
    m, n, k = 2 * bm + 5, 2 * bn + 24, 400
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=bm + bn)
    t = dga.tiling(m, n, k)
    t.m1, t.n1 = bm, bn
    t.wavesM = t.wavesN = 0      # the tile's own wave grid (a grid no build of the tile has is refused: dga_tiling_check)
    got = _run(dga, a, sfa, b, sfb, tiling=t)
    _check(oracle, got, a, sfa, b, sfb)


def test_random_bit_patterns_and_wild_scales(dga, oracle):
    """All e4m3fn encodings (subnormals, -0, max) and scales over many binades."""
    m, n, k = 192, 256, 640
    rng = np.random.default_rng(5)
    a = oracle.random_fp8_bytes((m, k), seed=1)
    b = oracle.random_fp8_bytes((n, k), seed=2)
    sfa = np.exp2(rng.uniform(-12, 4, size=(m, 5))).astype(np.float32)
    sfb = np.exp2(rng.uniform(-12, 4, size=(2, 5))).astype(np.float32)
    got = _run(dga, a, sfa, b, sfb)
    # arbitrary bit patterns span 15 binades: the hardware's worst-case alignment envelope applies
    _check(oracle, got, a, sfa, b, sfb, eps=2.0 ** -12, frac=1e-2)


def test_nan_bytes_propagate(dga, oracle):
    m, n, k = 64, 128, 256
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=9)
    a[3, 17] = 0x7F
    b[100, 200] = 0xFF
    got = _run(dga, a, sfa, b, sfb)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb)
    nan_g = (got & 0x7FFF) > 0x7F80
    nan_w = (want & 0x7FFF) > 0x7F80
    assert nan_w[3, :].all() and nan_w[:, 100].all() and nan_w.sum() == n + m - 1
    assert np.array_equal(nan_g, nan_w)
    oracle.assert_parity(got, want, a, sfa, b, sfb)


@pytest.mark.parametrize("m,n,k", [(40, 130, 100), (17, 33, 7), (5, 5, 129), (64, 128, 0)])
def test_generic_kernel_odd_k(dga, oracle, m, n, k):
    """K % 16 != 0 (and K = 0) take the element-wise HIP kernel."""
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=3)
    got = _run(dga, a, sfa, b, sfb)
    _check(oracle, got, a, sfa, b, sfb)


def test_empty_problem_is_a_noop(dga):
    out = torch.empty((0, 128), dtype=torch.bfloat16, device="cuda")
    a = torch.empty((0, 256), dtype=torch.uint8, device="cuda")
    sfa = torch.empty((0, 2), dtype=torch.float32, device="cuda")
    b = torch.zeros((128, 256), dtype=torch.uint8, device="cuda")
    sfb = torch.ones((1, 2), dtype=torch.float32, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True)


def test_linearity_in_scales_full_size(dga, oracle):
    """Size-independent property at BASELINE config 2 (4096^3), arbitrary bit patterns: doubling sfa doubles the fp32
    sum exactly, so the bf16 outputs differ by exactly one exponent step."""
    m = n = k = 4096
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 256, (m, k), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 256, (n, k), dtype=torch.uint8, device="cuda", generator=g)
    a = torch.where((a & 0x7F) == 0x7F, a & 0x80, a)
    b = torch.where((b & 0x7F) == 0x7F, b & 0x80, b)
    sfa = torch.rand((m, k // 128), device="cuda") + 0.5
    sfb = torch.rand((n // 128, k // 128), device="cuda") + 0.5
    o1 = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    o2 = torch.empty_like(o1)
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o1)
    dga.gemm_fp8_fp8_bf16_nt((a, sfa * 2), (b, sfb), o2, sync=True)
    assert torch.equal(o1.float() * 2, o2.float())


@pytest.mark.parametrize("workload", ["dense_4096", "dsv3_prefill"])
def test_baseline_configs_on_their_own_recipe(dga, oracle, workload):
    """BASELINE configs[1] (4096^3) and configs[2] (M=4096, K=7168, N=2048) on the inputs bench.py times -- amax-quantised
    normal data -- against the CPU oracle: 256 sampled rows, every column, eps = 2^-15, at most 2e-3 beyond 2 ulp."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    m, n, k = bench.WORKLOADS[workload]
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True)
    rows = np.arange(3, m, 16)[:256]
    an, san, bn, sbn = a.cpu().numpy(), sfa.cpu().numpy(), b.cpu().numpy(), sfb.cpu().numpy()
    want = oracle.gemm_fp8_fp8_bf16_nt(an[rows], san[rows], bn, sbn, threads=16)
    got = out[torch.from_numpy(rows).cuda()].view(torch.int16).cpu().numpy().view(np.uint16)
    oracle.assert_parity(got, want, an[rows], san[rows], bn, sbn, eps=2.0 ** -15, frac=2e-3)


@pytest.mark.parametrize("m,n,k,split", [(8, 1024, 4096, 4), (64, 512, 2048, 3), (100, 300, 1536, 5), (16, 128, 1024, 8)])
def test_split_k_parity(dga, oracle, m, n, k, split):
    """kernelSerial 4: K cut over `split` workgroups per tile, fp32 slabs in the workspace, combine kernel."""
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + n + k)
    t = dga.tiling(m, n, k)
    t.splitkFactor, t.kernelSerial = split, 4
    assert dga.workspace_bytes(t) >= split * m * n * 4
    got = _run(dga, a, sfa, b, sfb, tiling=t)
    _check(oracle, got, a, sfa, b, sfb)
    # the heuristic picks split-K by itself for a decode shape (up to 16 rows on a matrix of <= 10240 rows: the one-launch
    # workgroup split-K, kernelSerial 6 -- tests/test_wsk_gpu.py)
    th = dga.select_kernel(64, 7168, 18432)
    assert th.kernelSerial == 4 and th.splitkFactor > 1 and th.blockDim == -(-64 // th.m1) * (7168 // th.n1) * th.splitkFactor
    assert dga.select_kernel(8, 7168, 18432).kernelSerial == 6


def test_split_k_without_workspace_still_correct(dga, oracle):
    """The C ABI accepts workspace == NULL: the single-pass kernel runs instead (same answer)."""
    import ctypes
    from deepgemm_ascend_amd import _lib
    m, n, k = 16, 256, 2048
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=2)
    t = dga.tiling(m, n, k)
    t.splitkFactor, t.kernelSerial = 4, 4
    ta, tsa, tb, tsb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(ta.data_ptr(), tsa.data_ptr(), tb.data_ptr(), tsb.data_ptr(), out.data_ptr(),
                                             m, n, k, ctypes.byref(t), None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    _check(oracle, out.view(torch.int16).cpu().numpy().view(np.uint16), a, sfa, b, sfb)
    small = torch.empty(16, dtype=torch.uint8, device="cuda")   # a workspace that is passed must be big enough
    rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(ta.data_ptr(), tsa.data_ptr(), tb.data_ptr(), tsb.data_ptr(), out.data_ptr(),
                                             m, n, k, ctypes.byref(t), small.data_ptr(), 16, 0)
    assert rc == -7


@pytest.mark.parametrize("m,n,k", [(129, 257, 1001), (64, 384, 1921), (5, 130, 17)])
def test_odd_k_padding_pass(dga, oracle, m, n, k):
    """K % 16 != 0 with a workspace: operands are re-laid with zero-padded rows, then the LDS-DMA kernel runs."""
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=k)
    t = dga.tiling(m, n, k)
    assert dga.workspace_bytes(t) >= (m + n) * (-(-k // 128) * 128)
    got = _run(dga, a, sfa, b, sfb)
    _check(oracle, got, a, sfa, b, sfb)


@pytest.mark.parametrize("off_a,off_b", [(1, 3), (2, 0), (3, 2)])
def test_odd_k_padding_pass_takes_operands_at_any_byte_offset(dga, oracle, off_a, off_b):
    """The re-layout pass reads aligned dwords and shifts them into place: operands that start at odd byte addresses, rows of
    odd length, and a last row that ends on the last byte of its allocation."""
    m, n, k = 70, 200, 1001
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=off_a * 4 + off_b)
    buf_a = torch.zeros(off_a + m * k, dtype=torch.uint8, device="cuda")
    buf_b = torch.zeros(off_b + n * k, dtype=torch.uint8, device="cuda")
    ta = buf_a[off_a:].view(m, k); ta.copy_(_dev(a))
    tb = buf_b[off_b:].view(n, k); tb.copy_(_dev(b))
    assert ta.data_ptr() % 4 == off_a % 4
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, _dev(sfa)), (tb, _dev(sfb)), out, sync=True)
    _check(oracle, out.view(torch.int16).cpu().numpy().view(np.uint16), a, sfa, b, sfb)


def test_config3_full_size_properties(dga, oracle):
    """BASELINE config 3 (M=4096, K=7168, N=2048) at full size: (a) scaling a whole column block of sfb by 2 doubles
    exactly those 128 output columns, (b) row-permuting A permutes the output rows (bitwise), (c) 48 sampled rows
    against the oracle."""
    m, k, n = 4096, 7168, 2048
    import bench
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3)
    o1 = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o1)
    sfb2 = sfb.clone(); sfb2[5] *= 2
    o2 = torch.empty_like(o1)
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb2), o2)
    perm = torch.randperm(m, device="cuda")
    o3 = torch.empty_like(o1)
    dga.gemm_fp8_fp8_bf16_nt((a[perm].contiguous(), sfa[perm].contiguous()), (b, sfb), o3, sync=True)
    assert torch.equal(o2[:, 640:768].float(), o1[:, 640:768].float() * 2)
    assert torch.equal(torch.cat([o2[:, :640], o2[:, 768:]], 1), torch.cat([o1[:, :640], o1[:, 768:]], 1))
    assert torch.equal(o3, o1[perm])
    rows = np.arange(17, m, 85)
    want = oracle.gemm_fp8_fp8_bf16_nt(a.cpu().numpy()[rows], sfa.cpu().numpy()[rows], b.cpu().numpy(), sfb.cpu().numpy(), threads=16)
    got = o1[torch.from_numpy(rows).cuda()].view(torch.int16).cpu().numpy().view(np.uint16)
    oracle.assert_parity(got, want, a.cpu().numpy()[rows], sfa.cpu().numpy()[rows], b.cpu().numpy(), sfb.cpu().numpy())


def test_every_schedule_is_bitwise_identical(dga, oracle):
    """dispatchPolicyTag 0 / 1 / 2 (plain, ping-pong, continuous) order the same MFMAs and FMAs differently in time
    only: outputs must agree bit for bit."""
    m, n, k = 1000, 1300, 2176
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=8)
    outs = []
    for pol in (0, 1, 2):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.splitkFactor, t.kernelSerial = 256, 256, 2, 0, 0, 1, 0
        t.dispatchPolicyTag = pol
        outs.append(_run(dga, a, sfa, b, sfb, tiling=t))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    _check(oracle, outs[0], a, sfa, b, sfb)


def test_hip_graph_capture_and_replay(dga, oracle):
    """The launch path does no allocation or synchronisation: it can be captured into a HIP graph and replayed."""
    m, n, k = 512, 768, 1024
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=4)
    ta, tsa, tb, tsb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t, sync=True)   # warm: function attributes set outside capture
    out.zero_()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            dga.gemm_fp8_fp8_bf16_nt((ta, tsa), (tb, tsb), out, tiling_=t)
    torch.cuda.synchronize()
    assert (out == 0).all()          # capture does not execute
    g.replay(); torch.cuda.synchronize()
    _check(oracle, out.view(torch.int16).cpu().numpy().view(np.uint16), a, sfa, b, sfb)


@pytest.mark.parametrize("m,n,k", [(233, 1408, 5120), (1920, 512, 12928), (48, 3328, 512), (3789, 1280, 2176)])
def test_predictor_picks_run_and_match_the_oracle(dga, oracle, m, n, k):
    """Whatever the learned predictor selects must be a launchable build with the same parity as any other tiling."""
    dga.predictor_load(None)
    t, _, _ = dga.select_kernel_with_predictor(m, n, k)
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + n)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
                             (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()), out, tiling_=t, sync=True)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(out.view(torch.int16).cpu().numpy().view(np.uint16), want, a, sfa, b, sfb)


@pytest.mark.parametrize("m,n,k", [(4352, 4096, 128), (4300, 4100, 256), (4352, 4352, 144)])
def test_quarter_tile_tail(dga, oracle, m, n, k):
    """kernelSerial 5: whole waves of 256x256 tiles + the last partial wave in 128x128 tiles (second launch).  Same bytes
    as a single launch over all tiles; parity against the oracle on a sample of rows (incl. the tail region)."""
    import os
    t = dga.select_kernel(m, n, k)
    assert (t.m1, t.n1, t.kernelSerial) == (256, 256, 5), t.as_dict()
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=k)
    ta, tsfa, tb, tsfb = [torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb)]
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((ta, tsfa), (tb, tsfb), out, tiling_=t, sync=True)
    t0 = dga.select_kernel(m, n, k)
    t0.kernelSerial = 0
    ref = torch.zeros_like(out)
    dga.gemm_fp8_fp8_bf16_nt((ta, tsfa), (tb, tsfb), ref, tiling_=t0, sync=True)
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))
    rows = np.r_[0:64, m - 300:m]   # the head and the rows the tail tiles cover
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    oracle.assert_parity(out[torch.from_numpy(rows).cuda()].view(torch.int16).cpu().numpy().view(np.uint16), want,
                         a[rows], sfa[rows], b, sfb)


@pytest.mark.parametrize("m,n,k", [(256, 512, 256), (2048, 2048, 640), (5120, 4864, 896), (1024, 18432, 384), (300, 512, 256)])
def test_persistent_continuous_build_writes_the_same_bits(dga, m, n, k):
    """dispatchPolicyTag 6 (the 256x256 continuous pipeline, one workgroup per CU walking its tiles, the next tile's first
    k blocks fetched from inside the last ones) against the one-tile continuous build: identical output bytes -- one tile
    per workgroup, several (odd and even k-block counts, so the stage parity flips between tiles), and a shape with an M
    edge, which the launcher hands to the one-tile build."""
    gen = torch.Generator(device="cuda").manual_seed(m + n + k)
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen) | (
        torch.randint(0, 2, (m, k), dtype=torch.uint8, device="cuda", generator=gen) << 7)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    a[1, 3] = 0x7F                       # a NaN byte: its row must not leak into the next tile of the same workgroup
    kb, nb = -(-k // 128), -(-n // 128)
    sfa = torch.rand((m, kb), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((nb, kb), device="cuda", generator=gen) + 0.5
    outs = {}
    for pol in (2, 6):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 256, 256, 4, 2, 2, pol, 0, 1
        o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device="cuda")
        for _ in range(2):
            dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t, sync=True)
        outs[pol] = o
    assert torch.equal(outs[2].view(torch.int16), outs[6].view(torch.int16))
    nan_rows = torch.isnan(outs[6].float()).any(dim=1).nonzero().flatten().tolist()
    assert nan_rows == [1]


def test_persistent_continuous_with_the_quarter_tile_tail(dga):
    """A raster of 288 full tiles: the tiling asks for whole waves in the persistent continuous kernel (dispatchPolicyTag 6 on
    the first 256 tiles) plus the last partial wave in quarter tiles (kernelSerial 5); the bytes are those of one plain
    launch of the one-tile kernel."""
    m, n, k = 1024, 18432, 640
    gen = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((m, 5), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((n // 128, 5), device="cuda", generator=gen) + 0.5
    t = dga.select_kernel(m, n, k)
    assert (t.m1, t.n1, t.kernelSerial, t.dispatchPolicyTag) == (256, 256, 5, dga.api.POLICY_CONTINUOUS_PERSISTENT)
    o6 = torch.full((m, n), -1.0, dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o6, tiling_=t, sync=True)
    t2 = dga.select_kernel(m, n, k)
    t2.kernelSerial, t2.dispatchPolicyTag = 0, dga.api.POLICY_CONTINUOUS
    o2 = torch.full((m, n), -1.0, dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o2, tiling_=t2, sync=True)
    assert torch.equal(o6.view(torch.int16), o2.view(torch.int16))


@pytest.mark.parametrize("m,n,k", [(3584, 6400, 256), (3511, 6151, 272), (2816, 8192, 128)])
def test_quarter_tile_tail_longer_than_a_quarter_of_the_raster(dga, oracle, m, n, k):
    """A caller's tiling may name kernelSerial 5 on a raster whose partial last round is long (the selector stops at a quarter of the
    CUs): 350 parent tiles, 94 of them in the tail = 376 quarter tiles, MORE than the parent raster holds -- the kernel once read such
    an index as a group index and stored past the output (a GPU memory fault).  Same bytes as the single launch."""
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=k)
    ta, tsfa, tb, tsfb = [torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb)]
    outs = []
    for ks in (5, 0):
        t = dga.select_kernel(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.splitkFactor, t.kernelSerial = 256, 256, 0, 0, 2, 2, 1, ks
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((ta, tsfa), (tb, tsfb), out, tiling_=t, sync=True)
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    rows = np.r_[0:32, m - 200:m]
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    oracle.assert_parity(outs[0][torch.from_numpy(rows).cuda()].view(torch.int16).cpu().numpy().view(np.uint16), want, a[rows], sfa[rows], b, sfb)
