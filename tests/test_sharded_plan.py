"""CPU: the sharded forward's layout and plan come from the C ABI (dga_sharded_layout / dga_sharded_plan: pure host
arithmetic) -- the sizes are what the engine allocates, every event a step waits for has been recorded by an earlier step on
another stream, every chunk passes dispatch -> GEMM -> combine in that order, and the equal-split exchange sizes follow the
pair capacity.  (The plan's semantics run in tests/test_parallel_gloo.py: world-2 gloo, the plan interpreted on CPU tensors.)"""
import ctypes

import pytest

from deepgemm_ascend_amd import _lib


def _plan(**kw):
    base = dict(world=8, rank=3, groups_total=256, m_max=128, n=2048, k=7168, chunks=0, max_tokens=4096, capacity_factor=1.25,
                indexed=0, policy=-1)
    base.update(kw)
    sh = _lib.ShardedShape(*[base[f[0]] for f in _lib.ShardedShape._fields_])
    lay = _lib.ShardedLayout()
    L = _lib.lib()
    rc = L.dga_sharded_layout(ctypes.byref(sh), ctypes.byref(lay))
    if rc:
        return rc, None, None
    n = ctypes.c_int(0)
    assert L.dga_sharded_plan(ctypes.byref(sh), None, 0, ctypes.byref(n)) == 0 and n.value == lay.steps
    steps = (_lib.ShardedStep * n.value)()
    assert L.dga_sharded_plan(ctypes.byref(sh), steps, n.value, ctypes.byref(n)) == 0
    assert L.dga_sharded_plan(ctypes.byref(sh), steps, n.value - 1, ctypes.byref(n)) == -7      # DGA_E_WORKSPACE
    return 0, lay, list(steps)


def test_layout_of_baseline_config5():
    """BASELINE configs[4]: 256 experts over 8 ranks, 4096 tokens per rank, capacity factor 1.25."""
    rc, lay, steps = _plan()
    assert rc == 0
    assert (lay.groups_local, lay.chunks, lay.groups_per_chunk, lay.kb, lay.nb) == (32, 2, 16, 56, 16)
    assert lay.hdr_offset == 7168 + 224 and lay.row_bytes == 7424 and lay.row_bytes % 128 == 0
    assert lay.pair_capacity == 320                      # ceil(1.25 * 4096 / (8 * 2)) = 320, a multiple of 16
    assert lay.rows_per_chunk == 8 * 320 and lay.rows_total == 2 * 8 * 320
    assert lay.send_bytes == lay.recv_bytes == lay.rows_total * 7424
    assert lay.osend_bytes == lay.oback_bytes == lay.rows_total * 2048 * 2
    assert lay.packed_a_bytes == 32 * 128 * 7168 and lay.row_of_slot_bytes == 0 and lay.indexed == 0
    assert lay.events == 1 + 3 * 2 and lay.steps == len(steps)
    rc, lay_i, _ = _plan(indexed=1)
    assert lay_i.indexed == 1 and lay_i.packed_a_bytes == 0 and lay_i.row_of_slot_bytes == 32 * 128 * 8
    rc, lay_b, _ = _plan(capacity_factor=0.0)            # the provable bound: min(tokens, experts per chunk x m_max)
    assert lay_b.pair_capacity == min(4096, 16 * 128)


@pytest.mark.parametrize("kw", [dict(), dict(indexed=1), dict(chunks=1), dict(chunks=4, indexed=1), dict(world=2, rank=1, groups_total=8,
                                                                                                      m_max=64, n=256, k=512, max_tokens=128)])
def test_plan_is_well_ordered(kw):
    rc, lay, steps = _plan(**kw)
    assert rc == 0
    S = _lib
    recorded = {}
    seen = {c: [] for c in range(lay.chunks)}
    for i, st in enumerate(steps):
        assert 0 <= st.stream <= 2
        if st.op == S.STEP_RECORD_EVENT:
            assert st.event not in recorded and 0 <= st.event < lay.events
            recorded[st.event] = st.stream
        elif st.op == S.STEP_WAIT_EVENT:
            assert st.event in recorded and recorded[st.event] != st.stream, "an event is waited for before it is recorded"
        elif st.op in (S.STEP_ALL_TO_ALL_DISPATCH, S.STEP_ROUTE_RECEIVED, S.STEP_GEMM, S.STEP_ALL_TO_ALL_COMBINE):  # (+ the zero / copy steps)
            seen[st.chunk].append(st.op)
            assert st.row_begin == st.chunk * lay.rows_per_chunk and st.rows == lay.rows_per_chunk
        if st.op in (S.STEP_UNPACK, S.STEP_GATHER_OUT):
            assert not lay.indexed
    assert len(recorded) == lay.events
    for c, ops in seen.items():
        assert ops == [S.STEP_ALL_TO_ALL_DISPATCH, S.STEP_ROUTE_RECEIVED, S.STEP_GEMM, S.STEP_ALL_TO_ALL_COMBINE], (c, ops)
    # the three stages of different chunks sit on different streams: dispatch on 1, GEMM on 0, combine on 2
    by_op = {st.op: st.stream for st in steps}
    assert (by_op[S.STEP_ALL_TO_ALL_DISPATCH], by_op[S.STEP_GEMM], by_op[S.STEP_ALL_TO_ALL_COMBINE]) == (1, 0, 2)
    assert steps[-1].op == S.STEP_RESTORE_ORDER and steps[-1].stream == 0


def test_world1_has_no_exchange():
    rc, lay, steps = _plan(world=1, rank=0, indexed=1, max_tokens=0)
    ops = [s.op for s in steps]
    assert rc == 0 and lay.events == 0 and lay.rows_total == 0 and lay.max_tokens == 256 * 128
    assert ops == [_lib.STEP_ROUTE_SOURCE, _lib.STEP_ZERO_DROPPED, _lib.STEP_GEMM]
    rc, lay, steps = _plan(world=1, rank=0, indexed=0)
    assert [s.op for s in steps] == [_lib.STEP_ROUTE_SOURCE, _lib.STEP_ZERO_DROPPED, _lib.STEP_PACK, _lib.STEP_GEMM, _lib.STEP_RESTORE_ORDER]


def test_shape_errors_and_indexed_limits():
    assert _plan(groups_total=250)[0] == -2              # experts must divide over the ranks
    assert _plan(rank=8)[0] == -2
    assert _plan(chunks=3)[0] == -2                      # 32 experts per rank do not split into 3 chunks
    # K % 4 != 0 with more than one rank: the header / the scales inside a payload row would be misaligned -- refused by the
    # layout, not by a routing step halfway through a forward; a single rank has no payload rows and takes any K
    assert _plan(indexed=1, k=7170)[0] == -2 and _plan(indexed=0, k=130)[0] == -2
    rc, lay, _ = _plan(world=1, rank=0, indexed=0, k=130)
    assert rc == 0
    rc, lay, _ = _plan(indexed=1, max_tokens=400000, capacity_factor=0.0, m_max=16384)   # > 2 GiB of payload rows: 32-bit tile offsets
    assert rc == 0 and lay.indexed == 0
