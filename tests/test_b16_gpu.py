"""GPU: the framework's own entry points (run_mmad_rtc / run_mmad_bench), judged as the reference judges them:
fp32 numpy golden, np.isclose(rtol, atol=1e-9), mismatch fraction <= 1e-4
(/root/reference/deep_gemm_ascend/framework/tests/test.py:19-64, framework/benchmark/benchmark.py:20-22,384-398)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def heavy_tail(rng, shape):  # test.py:30-32
    return np.clip(rng.lognormal(mean=1.0, sigma=1.2, size=shape), 1, 10).astype(np.float32)


@pytest.mark.parametrize("batch,m,n,k", [(1, 1, 512, 128), (2, 96, 200, 320), (1, 257, 129, 100), (3, 16, 16, 16)])
def test_run_mmad_rtc_bf16(dga, oracle, batch, m, n, k):
    rng = np.random.default_rng(batch * 7 + m)
    x = torch.tensor(heavy_tail(rng, (batch, m, k))).to(torch.bfloat16)
    y = torch.tensor(heavy_tail(rng, (batch, k, n))).to(torch.bfloat16)
    golden = np.matmul(x.float().numpy(), y.float().numpy()).astype(np.float32)  # test.py:37 on the bf16-rounded inputs
    z = torch.zeros((batch, m, n), dtype=torch.float32, device="cuda")
    dga.run_mmad_rtc(x.cuda(), y.cuda(), z)
    ok, ratio = oracle.verify_isclose(z.cpu().numpy(), golden, rtol=2e-4)  # test.py:19
    assert ok, ratio
    # and against the oracle's restatement of the reference matmul (k-ascending fp32 sums), batch 0
    g0 = oracle.matmul_f32_nn(x[0].float().numpy(), y[0].float().numpy())
    assert oracle.verify_isclose(z[0].cpu().numpy(), g0, rtol=2e-4)[0]


def test_run_mmad_bench_fp16_and_param_writeback(dga, oracle):
    m, n, k = 96, 1536, 608
    rng = np.random.default_rng(3)
    x = torch.tensor(heavy_tail(rng, (m, k))).to(torch.float16)
    y = torch.tensor(heavy_tail(rng, (k, n))).to(torch.float16)
    golden = np.matmul(x.float().numpy(), y.float().numpy()).astype(np.float32)
    z = torch.empty((m, n), dtype=torch.float32, device="cuda")
    params = torch.tensor([1, 1, 3, 8, 20, 10] + [0] * 22, dtype=torch.int32, device="cuda")  # benchmark.py deepgemm_gemm
    dga.run_mmad_bench(x.cuda(), y.cuda(), z, params)
    ok, ratio = oracle.verify_isclose(z.cpu().numpy(), golden, rtol=1.5e-6 * 100)  # fp16 products are exact; sums differ by order
    assert ok, ratio
    p = params.cpu().tolist()
    assert p[:6] == [1, 1, 3, 8, 20, 10] and p[6:10] == [m, n, k, 1]       # gemm_bench.hpp:68-81
    assert p == dga.bench_params_fill(m, n, k, [1, 1, 3, 8, 20, 10])


def test_run_mmad_custom_is_a_noop(dga):
    z = torch.full((2, 4, 4), 7.0, device="cuda")
    dga.run_mmad_custom(torch.zeros((2, 4, 4), device="cuda"), torch.zeros((2, 4, 4), device="cuda"), z)
    assert (z == 7).all()   # include/impls/mmad.cpp:79 returns immediately


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("batch,m,n,k", [(1, 1024, 1024, 1024), (2, 300, 520, 200), (1, 4096, 512, 136), (3, 64, 64, 64),
                                          (1, 8, 1024, 4096), (1, 40, 640, 2048), (2, 20, 384, 1100), (1, 128, 512, 8192),
                                          (1, 2048, 2048, 512), (1, 4096, 2048, 192), (1, 4000, 2000, 200)])   # short tiles + split-K, 256x256 continuous, 128x256 3-stage
def test_tiled_16bit_path(dga, oracle, dtype, batch, m, n, k):
    """The workspace (tiled MFMA) path of run_mmad_rtc: transposing pre-pass + LDS-DMA kernel, all tails."""
    rng = np.random.default_rng(m + n + k)
    x = torch.tensor(heavy_tail(rng, (batch, m, k))).to(dtype)
    y = torch.tensor(heavy_tail(rng, (batch, k, n))).to(dtype)
    golden = np.matmul(x.float().numpy(), y.float().numpy()).astype(np.float32)
    z = torch.full((batch, m, n), float("nan"), dtype=torch.float32, device="cuda")
    dga.run_mmad_rtc(x.cuda(), y.cuda(), z)
    ok, ratio = oracle.verify_isclose(z.cpu().numpy(), golden, rtol=2e-4)
    assert ok, ratio


def test_16bit_path_without_workspace_matches(dga, oracle):
    import ctypes
    from deepgemm_ascend_amd import _lib
    rng = np.random.default_rng(0)
    m, n, k = 200, 264, 192
    x = torch.tensor(heavy_tail(rng, (m, k))).to(torch.bfloat16).cuda()
    y = torch.tensor(heavy_tail(rng, (k, n))).to(torch.bfloat16).cuda()
    z0 = torch.zeros((m, n), dtype=torch.float32, device="cuda"); z1 = torch.zeros_like(z0)
    rc = _lib.lib().dga_run_mmad_rtc(x.data_ptr(), y.data_ptr(), z0.data_ptr(), 1, m, n, k, _lib.DT_BF16, 0)
    assert rc == 0
    dga.run_mmad_rtc(x[None], y[None], z1[None])
    torch.cuda.synchronize()
    ok, ratio = oracle.verify_isclose(z1.cpu().numpy(), z0.cpu().numpy(), rtol=2e-5)
    assert ok, ratio
