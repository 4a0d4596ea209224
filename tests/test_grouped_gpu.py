"""GPU parity of the grouped masked-M path against the oracle (semantics: rows >= masked_m[g] untouched)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MAX_ULP = 2


def _grouped_inputs(oracle, g, mmax, n, k, seed):
    A, SFA, B, SFB = [], [], [], []
    for i in range(g):
        a, sfa, b, sfb = oracle.make_inputs(mmax, n, k, seed=seed * 1000 + i)
        A.append(a); SFA.append(sfa); B.append(b); SFB.append(sfb)
    return np.stack(A), np.stack(SFA), np.stack(B), np.stack(SFB)


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("g,mmax,n,k,masks", [
    (4, 16, 128, 256, [0, 1, 7, 16]),          # SURVEY.md 8c golden (iv)
    (6, 128, 256, 512, [128, 0, 64, 127, 1, 65]),
    (3, 200, 384, 384, [200, 129, 3]),          # m_max not a tile multiple, two m-tiles
    (5, 48, 136, 144, [48, 0, 0, 17, 33]),      # N tail with scalar stores, K tail chunk
])
def test_grouped_masked_parity(dga, oracle, g, mmax, n, k, masks):
    a, sfa, b, sfb = _grouped_inputs(oracle, g, mmax, n, k, seed=g + mmax)
    masked = np.array(masks, np.int32)
    sentinel = np.uint16(0x7FC1)  # a NaN pattern the kernel never produces: untouched rows keep it
    init = np.full((g, mmax, n), sentinel, np.uint16)
    out = torch.from_numpy(init.view(np.int16)).cuda().view(torch.bfloat16)
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked(
        (torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
        (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()),
        out, torch.from_numpy(masked).cuda(), expected_m=int(max(masks)), sync=True)
    got = _bits(out)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(a, sfa, b, sfb, init, masked, threads=8)
    for gi in range(g):
        mm = masks[gi]
        assert (got[gi, mm:] == sentinel).all(), f"group {gi}: rows >= masked_m were written"
        if mm:
            oracle.assert_parity(got[gi, :mm], want[gi, :mm], a[gi, :mm], sfa[gi, :mm], b[gi], sfb[gi])


def test_grouped_equals_dense_per_group(dga, oracle):
    """Size-independent property: a grouped call equals G dense calls (bitwise, same kernel arithmetic)."""
    g, mmax, n, k = 8, 128, 512, 1024
    a, sfa, b, sfb = _grouped_inputs(oracle, g, mmax, n, k, seed=77)
    ta, tsfa, tb, tsfb = [torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb)]
    masked = torch.full((g,), mmax, dtype=torch.int32, device="cuda")
    out = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ta, tsfa), (tb, tsfb), out, masked, expected_m=mmax)
    ref = torch.zeros_like(out)
    t = dga.tiling(mmax, n, k, groups=g, expected_m=mmax)
    for i in range(g):
        dga.gemm_fp8_fp8_bf16_nt((ta[i], tsfa[i]), (tb[i], tsfb[i]), ref[i], tiling_=None)
    torch.cuda.synchronize()
    # the tile shape changes neither the per-128-block MFMA nor the order of the fp32 promotion: bitwise equal
    assert torch.equal(out.view(torch.int16), ref.view(torch.int16))


def test_masked_m_out_of_range_is_clamped(dga, oracle):
    g, mmax, n, k = 2, 32, 128, 128
    a, sfa, b, sfb = _grouped_inputs(oracle, g, mmax, n, k, seed=5)
    masked = torch.tensor([1000, 5], dtype=torch.int32, device="cuda")
    out = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked(
        (torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
        (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()), out, masked, expected_m=32, sync=True)
    want = oracle.gemm_fp8_fp8_bf16_nt(a[0], sfa[0], b[0], sfb[0])
    oracle.assert_parity(_bits(out[0]), want, a[0], sfa[0], b[0], sfb[0])
    assert (out[1, 5:] == 0).all()


def test_copy_rows_gather_scatter(dga):
    g = torch.Generator(device="cuda").manual_seed(1)
    src = torch.randint(0, 256, (300, 7168 + 224), dtype=torch.uint8, device="cuda", generator=g)
    idx = torch.randperm(300, device="cuda", generator=g)[:200].contiguous()
    dst = torch.zeros((200, 7168), dtype=torch.uint8, device="cuda")
    dga.copy_rows(dst, src, src_index=idx, row_bytes=7168)                   # gather
    assert torch.equal(dst, src[idx, :7168])
    sc = torch.zeros((300, 224), dtype=torch.uint8, device="cuda")
    dga.copy_rows(sc, src, dst_index=idx, row_bytes=224, src_byte_offset=7168)  # scatter, offset source, 200 rows
    ref = torch.zeros_like(sc); ref[idx] = src[:200, 7168:]
    assert torch.equal(sc, ref)
    odd = torch.zeros((200, 37), dtype=torch.uint8, device="cuda")           # unaligned row bytes: scalar path
    dga.copy_rows(odd, src, src_index=idx, row_bytes=37)
    assert torch.equal(odd, src[idx, :37])


def test_expert_sharding_world1_on_gpu(dga, oracle):
    """The routing engine end to end on one GPU (no exchange): dispatch -> HIP grouped GEMM -> combine, every token's
    output row against the oracle's 1 x K row product."""
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    G, MMAX, N, K = 6, 32, 256, 512
    rng = np.random.default_rng(3)
    b = oracle.random_fp8_bytes((G, N, K), seed=1); sfb = rng.uniform(0.5, 1.5, (G, N // 128, K // 128)).astype(np.float32)
    T = 70
    ids = rng.integers(0, G, size=T); ids[ids == 4] = 5          # expert 4 stays empty
    q = oracle.random_fp8_bytes((T, K), seed=2); sf = rng.uniform(0.5, 1.5, (T, K // 128)).astype(np.float32)
    eng = ExpertShardedGroupedGemm(0, 1, G, MMAX, N, K, torch.device("cuda"), None)
    eng.set_weights(torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda())
    res = eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda())
    torch.cuda.synchronize()
    got = _bits(res)
    assert np.array_equal(eng.masked_m.cpu().numpy(), np.bincount(ids, minlength=G).astype(np.int32))
    for t in range(T):
        want = oracle.gemm_fp8_fp8_bf16_nt(q[t:t + 1], sf[t:t + 1], b[ids[t]], sfb[ids[t]])
        oracle.assert_parity(got[t:t + 1], want, q[t:t + 1], sf[t:t + 1], b[ids[t]], sfb[ids[t]], eps=2.0 ** -12, frac=1.0)


def test_config4_full_size(dga, oracle):
    """BASELINE config 4 at full size (256 experts x (M<=128, K=7168, N=2048), random masks): rows >= masked_m keep
    their sentinel everywhere; 6 sampled experts (incl. an empty one) against the oracle; and the grouped result of
    an expert equals the dense operator on that expert's rows (bitwise)."""
    G, MMAX, N, K = 256, 128, 2048, 7168
    g = torch.Generator(device="cuda").manual_seed(11)
    def rf(shape):
        x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
        x = torch.where((x & 0x7F) == 0x7F, x & 0x80, x)
        return torch.where((x & 0x78) > 0x60, x & 0xBF, x)     # keep magnitudes moderate
    a = rf((G, MMAX, K)); b = rf((G, N, K))
    sfa = torch.rand((G, MMAX, K // 128), device="cuda", generator=g) + 0.5
    sfb = torch.rand((G, N // 128, K // 128), device="cuda", generator=g) + 0.5
    masked = torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g)
    masked[7] = 0; masked[8] = 128; masked[9] = 1
    out = torch.full((G, MMAX, N), -7.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, expected_m=64, sync=True, policy="fast")
    mm = masked.cpu().numpy()
    row_ix = torch.arange(MMAX, device="cuda")[None, :, None]
    untouched = (row_ix >= masked[:, None, None])
    assert ((out == -7.0) | ~untouched).all()
    for e in (7, 8, 9, 100, 200, 255):
        r = int(mm[e])
        if r == 0:
            continue
        an, san, bn, sbn = a[e, :r].cpu().numpy(), sfa[e, :r].cpu().numpy(), b[e].cpu().numpy(), sfb[e].cpu().numpy()
        want = oracle.gemm_fp8_fp8_bf16_nt(an, san, bn, sbn, threads=16)
        oracle.assert_parity(_bits(out[e, :r]), want, an, san, bn, sbn, eps=2.0 ** -12, frac=1e-2)
        dense = torch.zeros((r, N), dtype=torch.bfloat16, device="cuda")
        t = dga.tiling(r, N, K); t.m1, t.n1, t.stages, t.splitkFactor, t.kernelSerial, t.wavesM, t.wavesN = 128, 256, 3, 1, 0, 2, 2
        dga.gemm_fp8_fp8_bf16_nt((a[e, :r].contiguous(), sfa[e, :r].contiguous()), (b[e], sfb[e]), dense, tiling_=t, sync=True)
        assert torch.equal(dense, out[e, :r])


def test_config4_full_size_on_the_survey_recipe(dga, oracle):
    """BASELINE configs[3] at full size on SURVEY.md 8(d)'s data recipe (fp32 ~ N(0,1), amax-quantised per 1x128 / 128x128
    by the product's quantisers), random masks.  Sampled experts against the CPU oracle: the fast path at its amax-quantised
    bar (2 ulp + 2^-15 S, at most 2e-3 of the elements beyond 2 ulp), the bf16-exact policy at 2^-22 S with at most 1e-5
    beyond 2 ulp, the strict policy bit for bit; rows >= masked_m keep their sentinel under every policy; and
    parallel.masked_parity (what bench.py prints as grouped.parity) agrees with the oracle's figures."""
    from deepgemm_ascend_amd import parallel
    G, MMAX, N, K = 256, 128, 2048, 7168
    g = torch.Generator(device="cuda").manual_seed(21)
    b, sfb = parallel._quantised_weights(G, N, K, g, "cuda")
    qa, qs = parallel._quantised_tokens(G * MMAX, K, g, "cuda")
    a, sfa = qa.view(G, MMAX, K), qs.view(G, MMAX, K // 128)
    masked = torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g)
    masked[7] = 0; masked[8] = 128; masked[9] = 1
    mm = masked.cpu().numpy()
    row_ix = torch.arange(MMAX, device="cuda")[None, :, None]
    untouched = (row_ix >= masked[:, None, None])
    sample = (7, 8, 9, 100, 200, 255)
    wants = {}
    for e in sample:
        r = int(mm[e])
        if r:
            wants[e] = oracle.gemm_fp8_fp8_bf16_nt(a[e, :r].cpu().numpy(), sfa[e, :r].cpu().numpy(), b[e].cpu().numpy(),
                                                   sfb[e].cpu().numpy(), threads=16)
    for policy, eps, frac in (("fast", 2.0 ** -15, 2e-3), ("bf16_exact", 2.0 ** -22, 1e-5), ("strict", 0.0, 0.0)):
        out = torch.full((G, MMAX, N), -7.0, dtype=torch.bfloat16, device="cuda")
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, expected_m=64, policy=policy, sync=True)
        assert ((out == -7.0) | ~untouched).all(), policy
        for e, want in wants.items():
            r = int(mm[e])
            got = _bits(out[e, :r])
            if policy == "strict":
                assert np.array_equal(got, want), f"expert {e}"
                continue
            rep = oracle.parity_report(got, want, a[e, :r].cpu().numpy(), sfa[e, :r].cpu().numpy(), b[e].cpu().numpy(),
                                       sfb[e].cpu().numpy())
            assert rep["nan_positions_equal"] and rep["worst_excess_over_S"] <= eps, (policy, e, rep)
            assert rep["frac_gt_max_ulp"] * got.size <= max(frac * got.size, 2), (policy, e, rep)
    it = torch.tensor(sample, device="cuda")
    rep = parallel.masked_parity(a[it].contiguous(), sfa[it].contiguous(), b[it].contiguous(), sfb[it].contiguous(), masked[it])
    assert rep["fast"]["masked_rows_untouched"] and rep["bf16_exact"]["masked_rows_untouched"]
    assert rep["fast"]["worst_excess_over_S"] <= 2.0 ** -15 * 1.01 and rep["fast"]["frac_gt_2ulp"] <= 2e-3, rep
    assert rep["bf16_exact"]["worst_excess_over_S"] <= 2.0 ** -22 * 1.01 and rep["bf16_exact"]["frac_gt_2ulp"] <= 1e-5, rep


def test_route_tokens(dga):
    """dga_route_tokens: counts = histogram, pos = a permutation that sorts the tokens by expert; out-of-range ids -> -1."""
    g = torch.Generator(device="cuda").manual_seed(3)
    for T, G in ((0, 8), (1, 1), (5000, 37), (32768, 256)):
        ids = torch.randint(0, G, (T,), device="cuda", generator=g)
        counts, pos = dga.route_tokens(ids, G)
        torch.cuda.synchronize()
        assert torch.equal(counts, torch.bincount(ids, minlength=G))
        if T:
            assert torch.equal(torch.sort(pos).values, torch.arange(T, device="cuda"))
            by_slot = torch.empty_like(ids); by_slot[pos] = ids
            assert (by_slot[1:] >= by_slot[:-1]).all()
    ids = torch.tensor([3, -1, 0, 9, 3], device="cuda")
    counts, pos = dga.route_tokens(ids, 4)
    assert counts.tolist() == [1, 0, 0, 2] and pos[1].item() == -1 and pos[3].item() == -1
    assert sorted(pos[[0, 2, 4]].tolist()) == [0, 1, 2] and pos[2].item() == 0


def test_copy_rows2(dga):
    """dga_copy_rows2: two row streams with shared indices, byte offsets inside rows, negative index = skip."""
    g = torch.Generator(device="cuda").manual_seed(5)
    rows, k, kb = 300, 256, 2
    q = torch.randint(0, 256, (rows, k), dtype=torch.uint8, device="cuda", generator=g)
    sf = torch.rand((rows, kb), device="cuda", generator=g)
    perm = torch.randperm(rows, device="cuda", generator=g)
    payload = torch.zeros((rows, k + 4 * kb), dtype=torch.uint8, device="cuda")
    dga.copy_rows2(payload, q, k, payload, sf.view(torch.uint8), 4 * kb, dst_index=perm, dst1_off=k)
    torch.cuda.synchronize()
    assert torch.equal(payload[perm][:, :k], q)
    assert torch.equal(payload[perm][:, k:].contiguous().view(torch.float32), sf)
    a = torch.zeros((rows, k), dtype=torch.uint8, device="cuda"); s2 = torch.zeros((rows, kb), device="cuda")
    idx = perm.clone(); idx[:7] = -1
    dga.copy_rows2(a, payload, k, s2.view(torch.uint8), payload, 4 * kb, src_index=idx, src1_off=k)
    torch.cuda.synchronize()
    assert torch.equal(a[7:], q[7:]) and torch.equal(s2[7:], sf[7:]) and (a[:7] == 0).all() and (s2[:7] == 0).all()


def test_loader_wave_build_writes_the_same_bits(dga, oracle):
    """dispatchPolicyTag 4 (extra waves that only issue the LDS-DMA; the tiling's choice for the masked weight stream)
    against the plain loop of the same tile: identical output bytes, ragged masks included."""
    g, mmax, n, k = 12, 128, 2048, 1024
    gen = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(0, 120, (g, mmax, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (g, n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((g, mmax, k // 128), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((g, n // 128, k // 128), device="cuda", generator=gen) + 0.5
    masked = torch.tensor([128, 0, 1, 64, 65, 127, 128, 33, 96, 128, 7, 128], dtype=torch.int32, device="cuda")
    t = dga.tiling(mmax, n, k, groups=g, expected_m=128)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = 128, 256, 2, 2, 3, 4
    big = dga.tiling(128, 2048, 7168, groups=256, expected_m=128)          # BASELINE configs[3]: the tiling's own choice
    assert (big.m1, big.n1, big.stages, big.dispatchPolicyTag) == (128, 256, 3, 5)   # 2048 tiles > CUs: the persistent form
    o4 = torch.full((g, mmax, n), -1.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o4, masked, 128, tiling_=t)
    t.dispatchPolicyTag = 0
    o0 = torch.full((g, mmax, n), -1.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o0, masked, 128, tiling_=t, sync=True)
    assert torch.equal(o4.view(torch.int16), o0.view(torch.int16))


@pytest.mark.parametrize("k", [400, 100])   # four k blocks with a K tail; a single (partial) k block per tile
@pytest.mark.parametrize("tile", [(128, 256, 2, 2), (128, 128, 2, 2), (64, 256, 1, 4), (64, 128, 1, 4), (16, 128, 1, 4)])
def test_persistent_build_writes_the_same_bits(dga, oracle, tile, k):
    """dispatchPolicyTag 5 (one workgroup per CU walks its tiles, the LDS ring runs across tile boundaries) against the
    one-tile plain loop of the same tile shape: identical output bytes.  The raster holds several tiles per workgroup
    (every workgroup crosses tile boundaries), empty experts in the middle of a workgroup's list, ragged masks, an N edge
    and a K tail; one expert is also checked against the oracle."""
    g, mmax, n = 300, 64, 392
    gen = torch.Generator(device="cuda").manual_seed(11)
    a = torch.randint(0, 120, (g, mmax, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (g, n, k), dtype=torch.uint8, device="cuda", generator=gen)
    kb, nb = -(-k // 128), -(-n // 128)
    sfa = torch.rand((g, mmax, kb), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((g, nb, kb), device="cuda", generator=gen) + 0.5
    masked = torch.randint(0, mmax + 1, (g,), dtype=torch.int32, device="cuda", generator=gen)
    masked[torch.rand((g,), device="cuda", generator=gen) < 0.3] = 0
    masked[5] = mmax
    outs = {}
    for pol in (0, 5):
        t = dga.tiling(mmax, n, k, groups=g, expected_m=mmax)
        t.m1, t.n1, t.wavesM, t.wavesN = tile
        t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 3, pol, 0, 1
        o = torch.full((g, mmax, n), -1.0, dtype=torch.bfloat16, device="cuda")
        for _ in range(2):   # a second launch on the same stream: nothing of the first one's ring state survives
            dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o, masked, mmax, tiling_=t, sync=True)
        outs[pol] = o
    assert torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
    an, sfan, bn, sfbn = a[5].cpu().numpy(), sfa[5].cpu().numpy(), b[5].cpu().numpy(), sfb[5].cpu().numpy()
    want = oracle.gemm_fp8_fp8_bf16_nt(an, sfan, bn, sfbn, threads=8)
    oracle.assert_parity(_bits(outs[5][5]), want, an, sfan, bn, sfbn, eps=2.0 ** -12, frac=1e-2)


def test_persistent_build_dense_and_contiguous(dga, oracle):
    """The persistent form on a dense raster (groups = 1, M and N edges) and on the contiguous layout (padding rows,
    padding tiles): the bytes of the plain loop."""
    gen = torch.Generator(device="cuda").manual_seed(12)
    m, n, k = 3000, 2100, 640
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((m, 5), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((-(-n // 128), 5), device="cuda", generator=gen) + 0.5
    outs = {}
    for pol in (0, 5):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 128, 128, 2, 2, 3, pol, 0, 1
        o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t, sync=True)
        outs[pol] = o
    assert torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
    groups, n, k = 6, 1024, 512
    seg = [256, 0, 128, 384, 128, 0]
    m = sum(seg) + 256                       # two padding tiles at the end
    idx = torch.full((m,), -1, dtype=torch.int32)
    pos = 0
    for gi, sg in enumerate(seg):
        idx[pos:pos + sg] = gi
        if sg:
            idx[pos + sg - 9:pos + sg] = -1   # padding rows behind a segment's valid rows
        pos += sg
    idx = idx.cuda()
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (groups, n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((m, 4), device="cuda", generator=gen) + 0.5
    sfb = torch.rand((groups, 8, 4), device="cuda", generator=gen) + 0.5
    outs = {}
    for pol in (0, 5):
        t = dga.tiling(m, n, k, groups=groups, contiguous=True)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 64, 128, 1, 4, 3, pol, 0, 1
        o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device="cuda")
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), o, idx, tiling_=t, sync=True)
        outs[pol] = o
    assert torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
    assert (outs[5][idx < 0] == -1.0).all()


@pytest.mark.parametrize("k", [128, 256, 640])   # one k block per tile, two (= refills in flight), many
def test_persistent_build_indexed_rows(dga, k):
    """The indexed form on the persistent build: every workgroup crosses tile boundaries, so the loader waves' row-table
    prefetch (one tile ahead, by LDS-DMA) is exercised -- including tiles shorter than the refill depth, where the wait in
    front of the prefetched entries is a full one; against the one-tile plain loop, bit for bit, and rows nobody owns
    stay untouched."""
    g, m_max, n = 96, 64, 1024
    kb = k // 128
    row_bytes = k + 4 * kb + 16
    rng = np.random.default_rng(21)
    masked = rng.integers(0, m_max + 1, size=g).astype(np.int32)
    masked[rng.random(g) < 0.25] = 0
    masked[0] = m_max
    rows = int(masked.sum()) + 13
    payload = rng.integers(0, 120, size=(rows, row_bytes), dtype=np.uint8)
    payload[:, k:k + 4 * kb] = rng.uniform(0.5, 1.5, size=(rows, kb)).astype(np.float32).view(np.uint8)
    perm = rng.permutation(rows)
    row_index = np.full((g, m_max), -1, np.int64)
    at = 0
    for i in range(g):
        row_index[i, :masked[i]] = perm[at:at + masked[i]]
        at += masked[i]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    tb = dev(rng.integers(0, 120, size=(g, n, k), dtype=np.uint8))
    tsfb = dev(rng.uniform(0.5, 1.5, size=(g, n // 128, kb)).astype(np.float32))
    tp, tidx, tm = dev(payload), dev(row_index.reshape(-1)), dev(masked)
    outs = {}
    for pol in (0, 5):
        t = dga.tiling(m_max, n, k, groups=g, expected_m=m_max)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 64, 128, 1, 4, 3, pol, 0, 1
        o = torch.full((rows, n), -5.0, dtype=torch.bfloat16, device="cuda")
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tp, tp, k, row_bytes // 4, (tb, tsfb), o, tidx, tm, m_max, m_max,
                                                          tiling_=t, sync=True)
        outs[pol] = o
    assert torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
    owned = np.zeros(rows, bool)
    owned[perm[:int(masked.sum())]] = True
    assert (outs[5][torch.from_numpy(~owned).cuda()] == -5.0).all()
    assert not (outs[5][torch.from_numpy(owned).cuda()] == -5.0).all()
