"""Host-side logic added in round 6 (CPU): arena ownership."""
import importlib

import pytest

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def ops():
    return importlib.import_module("6dgs_amd.ops")


def test_arena_serves_one_module_at_a_time(ops):
    """ops.Arena is a bump allocator that an IdentificationModule resets per scene: a reset by a SECOND live module would hand the first one's key planes out
    again under its feet.  reset(owner) claims the arena; another live owner is refused until release_owner() (or the first owner is gone)."""
    class Mod:            # stands in for an IdentificationModule (only identity and lifetime matter)
        pass
    a = ops.Arena(8 << 20, "cpu")
    m1, m2 = Mod(), Mod()
    a.take(1 << 20)
    a.reset(m1)
    assert a.mark() == 0
    a.take(1 << 20)
    a.reset(m1)                                  # the owner may reset as often as it likes
    with pytest.raises(RuntimeError, match="already serves another IdentificationModule"):
        a.reset(m2)
    a.reset()                                    # an anonymous reset (tools, tests) is the caller's responsibility, as before
    a.release_owner()
    a.reset(m2)
    del m2                                       # a dead owner no longer holds the arena
    import gc
    gc.collect()
    a.reset(m1)
