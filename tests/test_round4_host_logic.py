"""Host-side logic added in round 4 (CPU): how a batch is cut into launches of the select sweep -- by token row-count class in the Python layer
(IdentificationModule.score_tokens) and into launches of 8 images in the library (csrc/sweep_plan.h, instantiated for the host in hostcheck.cpp)."""
import ctypes as C
import importlib
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hc():
    b = importlib.import_module("6dgs_amd.build")
    lib = C.CDLL(b.build_hostcheck())
    lib.hc_sweep_launch_images.restype = C.c_int
    lib.hc_sweep_launch_images.argtypes = [C.c_int, C.c_int]
    return lib


def test_launch_classes_group_images_by_64_token_rows():
    im = importlib.import_module("6dgs_amd.identification_module")
    rows, order, cuts = im.launch_classes([256, 80, 129, 128, 1, 0, 193, 64, 65])
    assert rows == [4, 2, 3, 2, 1, 0, 4, 1, 2]
    assert order == [0, 6, 2, 1, 3, 8, 4, 7, 5]                    # descending rows, the caller's order inside a class
    assert cuts == [0, 2, 3, 6, 8, 9]
    groups = [order[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    assert all(len({rows[i] for i in g}) == 1 for g in groups) and sorted(sum(groups, [])) == list(range(9))
    assert im.launch_classes([256] * 5) == ([4] * 5, [0, 1, 2, 3, 4], [0, 5])      # RGB views: one class, the caller's order, no permutation
    assert im.launch_classes([]) == ([], [], [0, 0])


def test_sweep_launches_of_eight_with_a_tail_of_up_to_twelve(hc):
    for batch in range(1, 200):
        left, sizes = batch, []
        while left > 0:
            nb = hc.hc_sweep_launch_images(left, 8)
            assert 1 <= nb <= left
            sizes.append(nb)
            left -= nb
        assert sum(sizes) == batch and all(s == 8 for s in sizes[:-1])
        assert sizes[-1] <= 12 and (batch <= 12 or sizes[-1] >= 5)       # never a launch of one or two images behind full ones
        if batch <= 12:
            assert sizes == [batch]
    assert hc.hc_sweep_launch_images(77, 0) == 77 and hc.hc_sweep_launch_images(77, -1) == 77     # no cap
    assert [hc.hc_sweep_launch_images(n, 4) for n in (4, 6, 7, 9)] == [4, 6, 4, 4]
