"""GPU: two host threads, each on its own stream, issue split-K and odd-K GEMMs (both need a workspace) and grouped GEMMs at the
same time.  Workspaces are per (device, stream), the tiling cache and the kernel attributes are set up under locks: every result must
equal the one the same call gives alone."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_threads_two_streams(dga):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(9)
    cases = []
    for (m, n, k) in [(64, 1024, 8192), (48, 640, 1001), (300, 768, 2048), (16, 2048, 4096)]:
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device=dev, generator=g)
        b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device=dev, generator=g)
        sfa = torch.rand((m, -(-k // 128)), device=dev, generator=g) + 0.5
        sfb = torch.rand((-(-n // 128), -(-k // 128)), device=dev, generator=g) + 0.5
        want = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), want, sync=True)
        cases.append((a, sfa, b, sfb, want))
    errors = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(40):
                    a, sfa, b, sfb, want = cases[(it + seed) % len(cases)]
                    out = torch.empty_like(want)
                    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)
                    s.synchronize()
                    if not torch.equal(out.view(torch.int16), want.view(torch.int16)):
                        errors.append((seed, it))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]


def test_two_threads_two_streams_16_bit_operator(dga):
    """The 16-bit operator from two host threads on their own streams: the one-launch workgroup split-K, a deep small tile with split-K
    (per-stream workspace), the 8-wave 128x128 tile and a raster with a sub-tile tail (two launches) at the same time; every result
    equals the one the same call gives alone."""
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(11)
    cases = []
    for (m, n, k) in [(8, 2048, 2048), (64, 4096, 1024), (512, 1024, 512), (2100, 8000, 128), (24, 1536, 2048)]:
        x = (torch.randn((m, k), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        want = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
        dga.catlass_dynamic_matmul(x, w.t(), want, sync=True)
        cases.append((x, w, want))
    errors = []

    def worker(seed):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for it in range(40):
                    x, w, want = cases[(it + seed) % len(cases)]
                    out = torch.empty_like(want)
                    dga.catlass_dynamic_matmul(x, w.t(), out)
                    s.synchronize()
                    if not torch.equal(out.view(torch.int16), want.view(torch.int16)):
                        errors.append((seed, it))
        except Exception as e:   # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors[:5]
