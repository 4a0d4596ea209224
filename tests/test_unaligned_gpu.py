"""K % 16 != 0 WITHOUT the padded operand copies: the loader waves of the 128 x 256 tile fetch rows that start at any byte,
realign them in registers and write the LDS image themselves (csrc/gemm_fp8_kernel.hpp UNAL, kernelSerial 2 = the reference's
PaddingCommon kernel, which fuses its re-layout with the matmul:
/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_common_matmul_kernel.h:33-107).

Bars: bit identity with the padding-pass path (pad_rows + the same tile kernel on 16-byte aligned copies) and the fast path's bar
against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _run(dga, a, sfa, b, sfb, fused):
    m, k = a.shape
    n = b.shape[0]
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.splitkFactor = 128, 256, 3, 2, 2, 1
    t.dispatchPolicyTag = 4
    t.kernelSerial = 2 if fused else 0
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True, tiling_=t)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (130, 300, 1001), (128, 256, 129), (333, 520, 7681), (200, 260, 1036),    # K % 4 == 0 but K % 16 != 0
    (64, 256, 77), (5, 9, 15), (257, 513, 255), (1, 1, 1), (96, 1000, 2049),
])
def test_fused_padding_equals_the_padding_pass_and_the_oracle(dga, oracle, m, n, k):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + 3 * n + 5 * k)
    da, dsfa, db, dsfb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    got = _run(dga, da, dsfa, db, dsfb, True)
    ref = _run(dga, da, dsfa, db, dsfb, False)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ from the padding-pass path"
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(got, want, a, sfa, b, sfb, eps=oracle.eps_for_k(k))


def test_unaligned_base_pointers(dga, oracle):
    """Operands that start at odd addresses themselves (views into a larger buffer): still in place."""
    m, n, k = 70, 300, 333
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=1)
    buf_a = torch.zeros((m * k + 64,), dtype=torch.uint8, device="cuda"); buf_b = torch.zeros((n * k + 64,), dtype=torch.uint8, device="cuda")
    va = buf_a[7:7 + m * k].view(m, k); vb = buf_b[13:13 + n * k].view(n, k)
    va.copy_(_dev(a)); vb.copy_(_dev(b))
    got = _run(dga, va, _dev(sfa), vb, _dev(sfb), True)
    oracle.assert_parity(got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4), a, sfa, b, sfb)


@pytest.mark.parametrize("m,n,k", [(1279, 5003, 7681), (3511, 6151, 8191)])
def test_the_odd_shapes_of_the_reference_list_at_full_size(dga, m, n, k):
    """framework/benchmark/benchmark.py:24-44's odd shapes: every output equals the padding-pass path's."""
    gen = torch.Generator(device="cuda").manual_seed(k)
    a = torch.randint(0, 256, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 256, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    a[(a & 0x7F) == 0x7F] = 0x3C; b[(b & 0x7F) == 0x7F] = 0x3C
    kb = (k + 127) // 128
    sfa = torch.rand((m, kb), device="cuda", generator=gen) + 0.5
    sfb = torch.rand(((n + 127) // 128, kb), device="cuda", generator=gen) + 0.5
    got = _run(dga, a, sfa, b, sfb, True)
    ref = _run(dga, a, sfa, b, sfb, False)
    assert np.array_equal(got, ref)


def test_no_workspace_runs_in_place(dga, oracle):
    """Through the C ABI with workspace = NULL: odd K used to fall to the element-wise kernel; now the loader waves read it in place."""
    import ctypes
    from deepgemm_ascend_amd import _lib
    m, n, k = 150, 520, 1001
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=5)
    da, dsfa, db, dsfb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(ctypes.c_void_p(da.data_ptr()), ctypes.c_void_p(dsfa.data_ptr()), ctypes.c_void_p(db.data_ptr()),
                                             ctypes.c_void_p(dsfb.data_ptr()), ctypes.c_void_p(out.data_ptr()), m, n, k, None, None, 0,
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    oracle.assert_parity(_bits(out), oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4), a, sfa, b, sfb)
