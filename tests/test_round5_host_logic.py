"""Host-side logic added in round 5 (CPU): the arena the big per-scene buffers are carved from (6dgs_amd.ops.Arena) and the library's launch planner
as Python sees it (ops.select_sweep_plan -> sixdgs_select_sweep_plan: host arithmetic, no GPU)."""
import importlib

import pytest
import torch


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("6dgs_amd.ops")


def test_arena_bump_scopes_and_routing(ops):
    a = ops.Arena(64 << 20, "cpu")
    assert a.capacity == 64 << 20 and ops.get_arena() is None
    x = a.take(1000)
    y = a.take(5000)
    assert x.data_ptr() % 4096 == 0 and y.data_ptr() == x.data_ptr() + 4096 and a.off == (y.data_ptr() - a.buf.data_ptr()) + 5000      # 4 KB aligned addresses
    m = a.mark()
    z = a.take(1 << 20)
    assert z.data_ptr() % 4096 == 0 and a.high == a.off > m
    a.release(m)
    assert a.off == m and a.take(16).data_ptr() == z.data_ptr()                                                         # the released range is handed out again
    a.reset()
    assert a.off == 0 and a.high >= (1 << 20)
    with pytest.raises(RuntimeError, match="arena exhausted"):
        a.take(65 << 20)
    # routing: big buffers come from the installed arena, small ones and other devices from torch
    prev = ops.set_arena(a)
    try:
        assert prev is None and ops.get_arena() is a
        big = ops.big_empty((1 << 20, 6), torch.float32, "cpu")                       # 24 MB
        assert big.shape == (1 << 20, 6) and big.dtype == torch.float32 and 0 <= big.data_ptr() - a.buf.data_ptr() < 4096
        small = ops.big_empty(100, torch.uint8, "cpu")
        assert not (a.buf.data_ptr() <= small.data_ptr() < a.buf.data_ptr() + a.capacity)
        with ops.arena_scope():
            inner = ops.big_empty(8 << 20, torch.uint8, "cpu")
            assert inner.data_ptr() >= big.data_ptr() + big.numel() * 4
            with ops.arena_scope():
                ops.big_empty(8 << 20, torch.uint8, "cpu")
            assert a.off == (inner.data_ptr() - a.buf.data_ptr()) + (8 << 20)         # the inner scope gave its buffer back
        o0 = big.data_ptr() - a.buf.data_ptr()
        assert a.off == o0 + (24 << 20)                                               # ... and the outer one its own
        big[:] = 1.0
        assert float(a.buf[o0: o0 + (24 << 20)].view(torch.float32).sum()) == float(6 << 20)   # a view of the arena's bytes
    finally:
        ops.set_arena(None)
    with ops.arena_scope():                                                           # without an arena: plain torch tensors, the scope is a no-op
        t = ops.big_empty((4 << 20,), torch.uint8, "cpu")
    assert t.numel() == 4 << 20 and ops.get_arena() is None


def test_select_sweep_plan_through_the_library(ops):
    assert ops.select_sweep_plan([256] * 4) == [(4, 4)]
    assert ops.select_sweep_plan([64] * 32) == [(8, 32)]                              # four 64-token views per tile: ONE launch of 8 tiles
    assert ops.select_sweep_plan([192] * 4) == [(3, 4)]                               # 12 quarters = 3 full tiles
    assert ops.select_sweep_plan([40, 64, 65, 128, 200, 0, 256, 137]) == [(5, 8)]
    assert ops.select_sweep_plan(None, batch=20) == [(8, 8), (12, 12)]                # token counts unknown: one image per tile, 8 + a tail of 12
    assert ops.select_sweep_plan([0, 0]) == [(0, 2)] and ops.select_sweep_plan([]) == []
    assert ops.select_sweep_plan([256] * 13) == [(8, 8), (5, 5)]
