"""The `deep_gemm_cpp` torch extension (csrc/python_api_amd.cpp): the reference's pybind module name and its three entry
points (/root/reference/deep_gemm_ascend/framework/csrc/python_api.cpp:13-36) plus the fp8 operators, on the C ABI."""
import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def ext(dga):
    from deepgemm_ascend_amd import build_ext
    build_ext.build()
    from deepgemm_ascend_amd import deep_gemm_cpp
    return deep_gemm_cpp


def test_module_exports_the_reference_names(ext):
    for name in ("run_mmad_custom", "run_mmad_rtc", "run_mmad_bench",                      # python_api.cpp:33-35
                 "gemm_fp8_fp8_bf16_nt", "m_grouped_gemm_fp8_fp8_bf16_nt_masked", "m_grouped_gemm_fp8_fp8_bf16_nt_contiguous",
                 "per_token_cast_to_fp8", "per_block_cast_to_fp8", "get_m_alignment_for_contiguous_layout"):
        assert callable(getattr(ext, name)), name
    assert ext.abi_version() == 7 and ext.get_m_alignment_for_contiguous_layout() == 128


def test_cpu_tensors_are_refused(ext):
    a = torch.zeros((16, 128), dtype=torch.uint8); b = torch.zeros((128, 128), dtype=torch.uint8)
    with pytest.raises(RuntimeError):
        ext.gemm_fp8_fp8_bf16_nt(a, torch.ones((16, 1)), b, torch.ones((1, 1)), torch.zeros((16, 128), dtype=torch.bfloat16))


def test_operand_shapes_and_dtypes_are_checked_before_anything_is_launched(ext):
    """Every fp8 binding validates full shapes and dtypes (an undersized out / scale tensor would be an out-of-bounds
    device access); the checks run before the device check, so host tensors are enough to see them."""
    u8 = lambda *s: torch.zeros(s, dtype=torch.uint8)
    f32 = lambda *s: torch.ones(s, dtype=torch.float32)
    bf = lambda *s: torch.zeros(s, dtype=torch.bfloat16)
    dense = lambda **kw: ext.gemm_fp8_fp8_bf16_nt(kw.get("a", u8(16, 256)), kw.get("sfa", f32(16, 2)), kw.get("b", u8(128, 256)),
                                                  kw.get("sfb", f32(1, 2)), kw.get("out", bf(16, 128)), **kw.get("kw", {}))
    with pytest.raises(RuntimeError, match="out must be"):
        dense(out=bf(16, 64))
    with pytest.raises(RuntimeError, match="sfa must be"):
        dense(sfa=f32(16, 1))
    with pytest.raises(RuntimeError, match="sfb must be"):
        dense(sfb=f32(2, 2))
    with pytest.raises(RuntimeError, match="sfa must be"):
        dense(sfa=torch.ones((16, 2), dtype=torch.float64))
    with pytest.raises(RuntimeError, match="float8_e4m3fn or uint8"):
        dense(a=torch.zeros((16, 256), dtype=torch.int32))
    with pytest.raises(RuntimeError, match="out must be"):
        dense(out=torch.zeros((16, 128), dtype=torch.float16))
    with pytest.raises(RuntimeError, match="policy must be"):
        dense(kw={"policy": "exactish"})
    with pytest.raises(RuntimeError, match="contradicts"):
        dense(kw={"strict": True, "policy": "bf16_exact"})
    with pytest.raises(RuntimeError, match="HIP device"):   # everything else right: the device check is what is left
        dense()
    g, mmax, n, k = 2, 32, 128, 256
    masked = lambda **kw: ext.m_grouped_gemm_fp8_fp8_bf16_nt_masked(
        kw.get("a", u8(g, mmax, k)), kw.get("sfa", f32(g, mmax, 2)), kw.get("b", u8(g, n, k)), kw.get("sfb", f32(g, 1, 2)),
        kw.get("out", bf(g, mmax, n)), kw.get("masked_m", torch.zeros(g, dtype=torch.int32)), 16)
    with pytest.raises(RuntimeError, match="out must be"):
        masked(out=bf(g, mmax - 1, n))
    with pytest.raises(RuntimeError, match="sfa must be"):
        masked(sfa=f32(g, mmax, 1))
    with pytest.raises(RuntimeError, match="sfb must be"):
        masked(sfb=f32(g, 1, 1))
    with pytest.raises(RuntimeError, match="masked_m must be"):
        masked(masked_m=torch.zeros(g, dtype=torch.int64))
    with pytest.raises(RuntimeError, match="b must be"):
        masked(b=u8(g + 1, n, k))
    msum = 128
    contig = lambda **kw: ext.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(
        kw.get("a", u8(msum, k)), kw.get("sfa", f32(msum, 2)), kw.get("b", u8(g, n, k)), kw.get("sfb", f32(g, 1, 2)),
        kw.get("out", bf(msum, n)), kw.get("m_indices", torch.zeros(msum, dtype=torch.int32)))
    with pytest.raises(RuntimeError, match="out must be"):
        contig(out=bf(msum, n // 2))
    with pytest.raises(RuntimeError, match="m_indices must be"):
        contig(m_indices=torch.zeros(msum - 1, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="sfa must be"):
        contig(sfa=f32(msum // 2, 2))


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.gpu
def test_run_mmad_rtc_matches_the_golden_formula(ext, oracle):
    """test.py:23-38: golden = np.matmul(x1.astype(f32), x2.astype(f32)); the reference's bf16 tolerance (rtol 2e-4, 1e-4 budget)."""
    rng = np.random.default_rng(0)
    x = torch.from_numpy(rng.uniform(-1, 1, (2, 96, 320)).astype(np.float32)).to(torch.bfloat16).cuda()
    y = torch.from_numpy(rng.uniform(-1, 1, (2, 320, 128)).astype(np.float32)).to(torch.bfloat16).cuda()
    z = torch.zeros((2, 96, 128), dtype=torch.float32, device="cuda")
    ext.run_mmad_rtc(x, y, z)                                   # synchronous, output in place (gemm.hpp:110)
    golden = np.matmul(x.float().cpu().numpy(), y.float().cpu().numpy())
    ok, ratio = oracle.verify_isclose(z.cpu().numpy(), golden, rtol=2e-4)
    assert ok, ratio
    ext.run_mmad_custom(x, y, z)                                # a no-op, as the reference's (mmad.cpp:79)


@pytest.mark.gpu
def test_run_mmad_bench_writes_params_back(ext, dga):
    m, n, k = 96, 128, 256
    x = torch.randn((m, k), device="cuda").half(); y = torch.randn((k, n), device="cuda").half()
    z = torch.zeros((m, n), dtype=torch.float32, device="cuda")
    params = torch.zeros(28, dtype=torch.int32, device="cuda")
    params[:6] = torch.tensor([1, 1, 3, 8, 20, 10], dtype=torch.int32)
    ext.run_mmad_bench(x, y, z, params)
    assert params.cpu().tolist() == dga.bench_params_fill(m, n, k, [1, 1, 3, 8, 20, 10])   # gemm_bench.hpp:68-81
    assert torch.allclose(z, x.float() @ y.float(), rtol=2e-3, atol=1e-2)


@pytest.mark.gpu
def test_bf16_exact_policy_through_the_binding(ext, oracle):
    m, n, k = 200, 392, 912
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=1)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    ext.gemm_fp8_fp8_bf16_nt(dev(a), dev(sfa), dev(b), dev(sfb), out, policy="bf16_exact")
    torch.cuda.synchronize()
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4)
    rep = oracle.parity_report(_bits(out), want, a, sfa, b, sfb)
    assert rep["max_ulp"] <= 2 or rep["frac_gt_max_ulp"] * out.numel() <= 2, rep
    assert rep["worst_excess_over_S"] <= 2.0 ** -22, rep


@pytest.mark.gpu
@pytest.mark.parametrize("strict", [False, True])
def test_fp8_operators_against_the_oracle(ext, oracle, strict):
    m, n, k = 200, 392, 912
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=1)
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    ext.gemm_fp8_fp8_bf16_nt(dev(a), dev(sfa), dev(b), dev(sfb), out, strict=strict)
    torch.cuda.synchronize()
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4)
    if strict:
        assert np.array_equal(_bits(out), want)
    else:
        oracle.assert_parity(_bits(out), want, a, sfa, b, sfb)
    # grouped masked
    g, mmax, n, k = 4, 64, 256, 384
    parts = [oracle.make_inputs(mmax, n, k, seed=10 + i) for i in range(g)]
    A, SFA, B, SFB = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([0, 1, 33, 64], np.int32)
    outg = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
    ext.m_grouped_gemm_fp8_fp8_bf16_nt_masked(dev(A), dev(SFA), dev(B), dev(SFB), outg, dev(masked), 64, strict=strict)
    torch.cuda.synchronize()
    wantg = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(A, SFA, B, SFB, np.zeros((g, mmax, n), np.uint16), masked)
    gotg = _bits(outg)
    for i in range(g):
        mm = int(masked[i])
        assert (gotg[i, mm:] == 0).all()
        if mm and strict:
            assert np.array_equal(gotg[i, :mm], wantg[i, :mm])
        elif mm:
            oracle.assert_parity(gotg[i, :mm], wantg[i, :mm], A[i, :mm], SFA[i, :mm], B[i], SFB[i])
    # quantiser: byte-exact against the oracle's
    x = torch.randn((64, 512), device="cuda")
    q, sf = ext.per_token_cast_to_fp8(x)
    wq, wsf = oracle.quant_1x128(x.cpu().numpy())
    assert np.array_equal(q.view(torch.uint8).cpu().numpy(), wq) and np.array_equal(sf.cpu().numpy(), wsf)
