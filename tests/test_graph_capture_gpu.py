"""GPU: every asynchronous entry point can be captured into one HIP graph and replayed (no allocation, no read-back, no
synchronisation on the launch path) -- the way a serving layer issues kernels too short to be launched one by one
(INTEGRATION.md "Host cost of a call").  One graph holds: the activation quantiser, the dense GEMM with split-K (two launches and a
workspace), an odd-K dense GEMM (the re-layout pass), the masked grouped GEMM, the contiguous grouped GEMM, the bf16-exact and
strict policies and the aclnn operator's 16-bit path; the replayed results equal the eager ones bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_every_entry_point_in_one_hip_graph(dga):
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn((256, 1024), device=dev, generator=g, dtype=torch.bfloat16)
    w = torch.randn((512, 1024), device=dev, generator=g)
    wq, wsf = dga.per_block_cast_to_fp8(w)
    xo = torch.randn((64, 1001), device=dev, generator=g)           # odd K
    aq_o, asf_o = dga.per_token_cast_to_fp8(torch.nn.functional.pad(xo, (0, 23)))
    aq_o = aq_o.view(torch.uint8)[:, :1001].contiguous(); asf_o = asf_o.contiguous()
    bq_o = torch.randint(0, 120, (256, 1001), dtype=torch.uint8, device=dev, generator=g)
    bsf_o = torch.rand((2, 8), device=dev, generator=g) + 0.5
    G, MM = 4, 64
    ga = torch.randint(0, 120, (G, MM, 1024), dtype=torch.uint8, device=dev, generator=g)
    gsa = torch.rand((G, MM, 8), device=dev, generator=g) + 0.5
    gb = torch.randint(0, 120, (G, 512, 1024), dtype=torch.uint8, device=dev, generator=g)
    gsb = torch.rand((G, 4, 8), device=dev, generator=g) + 0.5
    masked = torch.tensor([64, 0, 17, 33], dtype=torch.int32, device=dev)
    idx = torch.arange(G, dtype=torch.int32, device=dev).repeat_interleave(128).contiguous()
    ca = torch.randint(0, 120, (G * 128, 1024), dtype=torch.uint8, device=dev, generator=g)
    csa = torch.rand((G * 128, 8), device=dev, generator=g) + 0.5
    h = torch.randn((96, 320), device=dev, generator=g, dtype=torch.float16)
    hw = torch.randn((160, 320), device=dev, generator=g, dtype=torch.float16)
    t_split = dga.tiling(256, 512, 1024)
    t_split.m1, t_split.n1, t_split.stages, t_split.splitkFactor, t_split.kernelSerial, t_split.dispatchPolicyTag = 64, 128, 3, 2, 4, 0

    outs = {k: torch.zeros(s, dtype=d, device=dev) for k, (s, d) in {
        "dense": ((256, 512), torch.bfloat16), "bx": ((256, 512), torch.bfloat16), "strict": ((256, 512), torch.bfloat16),
        "odd": ((64, 256), torch.bfloat16), "masked": ((G, MM, 512), torch.bfloat16), "contig": ((G * 128, 512), torch.bfloat16),
        "op16": ((96, 160), torch.float16)}.items()}
    q = torch.zeros((256, 1024), dtype=torch.uint8, device=dev)
    sf = torch.zeros((256, 8), dtype=torch.float32, device=dev)

    def layer():
        qq, ss = dga.per_token_cast_to_fp8(x)      # allocates its outputs: inside a capture they come from the graph's pool
        q.copy_(qq.view(torch.uint8)); sf.copy_(ss)
        dga.gemm_fp8_fp8_bf16_nt((q, sf), (wq, wsf), outs["dense"], tiling_=t_split)
        dga.gemm_fp8_fp8_bf16_nt((q, sf), (wq, wsf), outs["bx"], policy="bf16_exact")
        dga.gemm_fp8_fp8_bf16_nt((q, sf), (wq, wsf), outs["strict"], strict=True)
        dga.gemm_fp8_fp8_bf16_nt((aq_o, asf_o), (bq_o, bsf_o), outs["odd"])
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ga, gsa), (gb, gsb), outs["masked"], masked, 32)
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((ca, csa), (gb, gsb), outs["contig"], idx)
        dga.catlass_dynamic_matmul(h, hw.t(), outs["op16"])

    layer(); torch.cuda.synchronize()              # eager: the reference results, and every workspace / function attribute exists
    want = {k: v.clone() for k, v in outs.items()}
    want_q = q.clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        layer()                                    # the capture stream's own workspaces
    side.synchronize()
    for v in outs.values():
        v.zero_()
    q.zero_()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        layer()
    torch.cuda.synchronize()
    assert all(int((v != 0).sum()) == 0 for v in outs.values())     # capture executes nothing
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(q, want_q)
    for k, v in outs.items():
        assert torch.equal(v.view(torch.int16), want[k].view(torch.int16)), k


def test_the_16_bit_operators_plans_in_one_hip_graph(dga):
    """The 16-bit operator's round-4 kernels under capture: the one-launch workgroup split-K (one and two row tiles), a deep small
    tile with split-K (tile kernel + combine), the 128x128 8-wave build and a raster whose last partial round runs in sub-tiles
    (two launches, forced by $DGA_B16_PLAN, which is read per call); replay == eager, bit for bit."""
    import os
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(5)
    cases = {"wsk8": (8, 2048, 1024, None), "wsk24": (24, 1536, 2048, None), "deep": (64, 4096, 1024, "64,128,4"),
             "w8": (512, 1024, 512, "128,128,1,0,1"), "tail": (2100, 8000, 128, "256,256,1,64")}
    data = {}
    for name, (m, n, k, plan) in cases.items():
        x = (torch.randn((m, k), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        w = (torch.randn((n, k), device=dev, generator=g) * 0.5).to(torch.bfloat16)
        data[name] = (x, w, torch.zeros((m, n), dtype=torch.bfloat16, device=dev), plan)

    def layer():
        for name, (x, w, o, plan) in data.items():
            if plan:
                os.environ["DGA_B16_PLAN"] = plan
            try:
                dga.catlass_dynamic_matmul(x, w.t(), o)
            finally:
                os.environ.pop("DGA_B16_PLAN", None)

    layer(); torch.cuda.synchronize()
    want = {k: v[2].clone() for k, v in data.items()}
    for name, (x, w, o, _) in data.items():
        ref = x.float() @ w.float().t()
        assert bool(((o.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all()), name
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        layer()
    side.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        layer()
    for v in data.values():
        v[2].zero_()
    graph.replay(); torch.cuda.synchronize()
    for name, v in data.items():
        assert torch.equal(v[2].view(torch.int16), want[name].view(torch.int16)), name
