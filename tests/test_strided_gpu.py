"""Row-strided operands of the dense fp8 GEMM (dga_gemm_fp8_fp8_bf16_nt_strided) and the quantisers' aligned-row forms
(dga_cast_to_fp8_*_ld).

The reference's Python entry takes torch tensors with whatever strides they have
(/root/reference/deep_gemm_ascend/framework/csrc/python_api.cpp:18) and fuses the re-layout of unaligned operands into the matmul
launch (aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_common_matmul_kernel.h:33-107).  Here rows that start on 16-byte
boundaries are read where they lie; K % 16 != 0 in place needs zero row tails (the `_ld` quantisers write them), and an operand
without that promise goes through the padding pass alone.  Every variant must give the BYTES of the contiguous call (same tile
kernel, same zeros), and the oracle's within the policy's bar.
"""
import ctypes

import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev, EPS_ARBITRARY

pytestmark = pytest.mark.gpu


def _run(dga, a, sfa, b, sfb, m, n, **kw):
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True, **kw)
    return _bits(out)


def _padded_view(x, ld, fill):
    """[rows, k] view of a [rows, ld] device buffer: x in the first k bytes, zeros up to the next 16-byte boundary, `fill` after"""
    rows, k = x.shape
    buf = torch.full((rows, ld), fill, dtype=torch.uint8, device="cuda")
    buf[:, :k] = _dev(x)
    buf[:, k:(k + 15) // 16 * 16] = 0
    return buf[:, :k]


POLICIES = [{}, {"policy": "bf16_exact"}, {"strict": True}]


@pytest.mark.parametrize("kw", POLICIES, ids=["fast", "bf16_exact", "strict"])
@pytest.mark.parametrize("m,n,k,lda,ldb", [
    (256, 384, 512, 640, 528),        # K % 128 == 0, views of wider buffers
    (100, 300, 1168, 1168 + 16, 1168 + 256),   # K % 128 == 16
    (64, 4096, 1000, 1008, 1008),     # K % 16 == 8, rows round_up(K, 16) apart; split-K territory
    (333, 520, 777, 784, 800),        # K odd
    (1, 8, 5, 16, 16),                # less than one chunk
    (130, 257, 7681, 7696, 7696),     # the reference list's K
    (8, 520, 2056, 2064, 2176),       # decode rows: the selector's one-launch workgroup split-K reads the strided rows too
    (16, 4100, 4096, 4096 + 128, 4096 + 16),
])
def test_strided_operands_give_the_contiguous_call_s_bytes(dga, oracle, m, n, k, lda, ldb, kw):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + 3 * n + 7 * k)
    ref = _run(dga, _dev(a), _dev(sfa), _dev(b), _dev(sfb), m, n, **kw)
    av, bv = _padded_view(a, lda, 0x55), _padded_view(b, ldb, 0x7E)    # (0x7E = 448: garbage past the boundary must not be read)
    assert av.stride(0) == lda and bv.stride(0) == ldb and (m == 1 or not av.is_contiguous())
    got = _run(dga, av, _dev(sfa), bv, _dev(sfb), m, n, zero_padded=(True, True), **kw)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ"
    # without the promise the padding pass re-lays the strided rows out: same bytes again
    got2 = _run(dga, av, _dev(sfa), bv, _dev(sfb), m, n, **kw)
    assert np.array_equal(got2, ref)
    # one operand each way (weights padded at load time, activations as they come)
    got3 = _run(dga, _dev(a), _dev(sfa), bv, _dev(sfb), m, n, zero_padded=(False, True), **kw)
    got4 = _run(dga, av, _dev(sfa), _dev(b), _dev(sfb), m, n, zero_padded=(True, False), **kw)
    assert np.array_equal(got3, ref) and np.array_equal(got4, ref)
    if kw.get("strict"):
        assert np.array_equal(ref, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8))


def test_nonzero_tail_without_the_promise_is_not_read(dga, oracle):
    """Rows 16-byte aligned but with garbage right behind byte K: without zero_padded the call must not read them in place."""
    m, n, k, ld = 96, 256, 1000, 1008
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=5)
    ref = _run(dga, _dev(a), _dev(sfa), _dev(b), _dev(sfb), m, n)
    abuf = torch.full((m, ld), 0x7E, dtype=torch.uint8, device="cuda"); abuf[:, :k] = _dev(a)
    bbuf = torch.full((n, ld), 0x7E, dtype=torch.uint8, device="cuda"); bbuf[:, :k] = _dev(b)
    got = _run(dga, abuf[:, :k], _dev(sfa), bbuf[:, :k], _dev(sfb), m, n)
    assert np.array_equal(got, ref)


def test_bad_strides_are_refused(dga):
    from deepgemm_ascend_amd import _lib
    L = _lib.lib()
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    sf = torch.ones(64, dtype=torch.float32, device="cuda")
    out = torch.zeros((16, 128), dtype=torch.bfloat16, device="cuda")
    call = lambda lda, ldb, k=200: L.dga_gemm_fp8_fp8_bf16_nt_strided(buf.data_ptr(), lda, sf.data_ptr(), buf.data_ptr(), ldb, sf.data_ptr(),
                                                                      out.data_ptr(), 16, 128, k, 0, None, None, 0, None)
    E_SHAPE, E_ALIGN = -2, -4    # include/dga_hip.h
    assert call(199, 208) == E_SHAPE and call(208, 100) == E_SHAPE
    assert call(204, 208) == E_ALIGN and call(208, 212) == E_ALIGN
    assert call(200, 208) == 0 and call(208, 200) == 0    # a stride of exactly K is the contiguous layout
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,k", [(64, 512), (130, 1000), (17, 77), (200, 7681), (5, 3)])
def test_aligned_row_quantisers(dga, oracle, dtype, rows, k):
    g = torch.Generator(device="cuda").manual_seed(rows + k)
    x = (torch.randn((rows, k), device="cuda", generator=g) * 2.0).to(dtype)
    for fn in (dga.per_token_cast_to_fp8, dga.per_block_cast_to_fp8):
        q0, sf0 = fn(x)
        q1, sf1 = fn(x, aligned_rows=True)
        torch.cuda.synchronize()
        ld = (k + 127) // 128 * 128
        assert tuple(q1.shape) == (rows, k) and torch.equal(sf0, sf1)
        assert torch.equal(q0.view(torch.uint8), q1.view(torch.uint8))
        if k % 128:
            assert q1.stride(0) == ld and getattr(q1, "_dga_zero_padded", False)
            whole = torch.as_strided(q1.view(torch.uint8), (rows, ld), (ld, 1))
            assert int(whole[:, k:].max()) == 0, "row tails must be zero"
        else:
            assert q1.is_contiguous()


@pytest.mark.parametrize("k,ldq", [(1000, 1008), (77, 80), (77, 128), (7681, 7696), (512, 512)])
def test_ld_quantisers_through_the_c_abi(dga, k, ldq):
    """Any row stride from K to the end of the last 128-wide block (here: the 16-byte minimum); strides outside are refused."""
    from deepgemm_ascend_amd import _lib
    L = _lib.lib()
    rows = 37
    x = torch.randn((rows, k), device="cuda", generator=torch.Generator(device="cuda").manual_seed(k))
    kb = (k + 127) // 128
    for name, sf_rows in (("dga_cast_to_fp8_1x128", rows), ("dga_cast_to_fp8_128x128", 1)):
        q0 = torch.empty((rows, k), dtype=torch.uint8, device="cuda"); s0 = torch.empty((sf_rows, kb), device="cuda")
        q1 = torch.full((rows, ldq), 0xAA, dtype=torch.uint8, device="cuda"); s1 = torch.empty((sf_rows, kb), device="cuda")
        assert getattr(L, name)(x.data_ptr(), _lib.DT_FP32, rows, k, q0.data_ptr(), s0.data_ptr(), None) == 0
        assert getattr(L, name + "_ld")(x.data_ptr(), _lib.DT_FP32, rows, k, q1.data_ptr(), ldq, s1.data_ptr(), None) == 0
        torch.cuda.synchronize()
        assert torch.equal(q1[:, :k], q0) and torch.equal(s0, s1)
        assert ldq == k or int(q1[:, k:].max()) == 0
        assert getattr(L, name + "_ld")(x.data_ptr(), _lib.DT_FP32, rows, k, q1.data_ptr(), k - 1, s1.data_ptr(), None) == -2
        assert getattr(L, name + "_ld")(x.data_ptr(), _lib.DT_FP32, rows, k, q1.data_ptr(), kb * 128 + 16, s1.data_ptr(), None) == -2


@pytest.mark.parametrize("kw", POLICIES[:2], ids=["fast", "bf16_exact"])
def test_quantise_then_multiply_without_a_padding_pass(dga, oracle, kw):
    """bf16 activations and weights with K = 7681 -> aligned-row quantisers -> GEMM in place: the bytes of the contiguous pipeline,
    and no pad_rows launch (checked through the workspace: none is needed, so none is passed)."""
    from deepgemm_ascend_amd import _lib
    m, n, k = 200, 520, 7681
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((m, k), device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    (qa, sa), (qb, sb) = dga.per_token_cast_to_fp8(x), dga.per_block_cast_to_fp8(w)
    ref = _run(dga, qa, sa, qb, sb, m, n, **kw)
    (pa, sa1), (pb, sb1) = dga.per_token_cast_to_fp8(x, aligned_rows=True), dga.per_block_cast_to_fp8(w, aligned_rows=True)
    got = _run(dga, pa, sa1, pb, sb1, m, n, **kw)
    assert np.array_equal(got, ref)
    # the same call through the C ABI with NO workspace: in place or not at all (a padding pass would need one; the element-wise
    # fallback would give other bytes on cancellation-heavy outputs and take ~100x longer)
    t = dga.tiling(m, n, k)
    t.splitkFactor, t.kernelSerial = 1, 0
    if kw:
        t.dispatchPolicyTag = 7
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt_strided(pa.data_ptr(), pa.stride(0), sa1.data_ptr(), pb.data_ptr(), pb.stride(0), sb1.data_ptr(),
                                                     out.data_ptr(), m, n, k, 3, ctypes.byref(t), None, 0, None)
    assert rc == 0
    torch.cuda.synchronize()
    a_np, b_np = qa.view(torch.uint8).cpu().numpy(), qb.view(torch.uint8).cpu().numpy()
    want = oracle.gemm_fp8_fp8_bf16_nt(a_np, sa.cpu().numpy(), b_np, sb.cpu().numpy(), threads=8)
    if kw:
        _assert_bar(oracle, _bits(out), want, a_np, sa.cpu().numpy(), b_np, sb.cpu().numpy(), eps=EPS_ARBITRARY)
    else:
        oracle.assert_parity(_bits(out), want, a_np, sa.cpu().numpy(), b_np, sb.cpu().numpy())
