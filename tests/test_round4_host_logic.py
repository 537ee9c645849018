"""Host-side logic of the select sweep's launches (CPU; csrc/sweep_plan.h, instantiated for the host in hostcheck.cpp): launches of 8 tiles with a tail
of up to 12 (round 4) and the packing of a launch's images into 256-token tiles by their token counts (round 5)."""
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
    lib.hc_sweep_pack.restype = C.c_int
    lib.hc_sweep_pack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
    return lib


def pack(hc, n_tok, cap=8, batch=None):
    """sweep_pack -> list of launches: dict(n_slots, n_images, q_img [S][4], q_lq [S][4], img, img_nq, img_q [n][4])."""
    import numpy as np
    b = len(n_tok) if n_tok is not None else batch
    per, S = hc.hc_sweep_pack_ints(), hc.hc_sweep_max_slots()
    out = np.zeros((max(b, 1) + 1, per), np.int32)
    arr = (C.c_int * b)(*n_tok) if n_tok is not None else None
    n = hc.hc_sweep_pack(arr, b, cap, out.ctypes.data, out.shape[0])
    assert 0 <= n <= out.shape[0]
    res = []
    for row in out[:n]:
        ns, ni = int(row[0]), int(row[1])
        o = 2
        q_img = row[o:o + 4 * S].reshape(S, 4); o += 4 * S
        q_lq = row[o:o + 4 * S].reshape(S, 4); o += 4 * S
        img = row[o:o + 4 * S]; o += 4 * S
        img_nq = row[o:o + 4 * S]; o += 4 * S
        img_q = row[o:o + 16 * S].reshape(4 * S, 4)
        res.append(dict(n_slots=ns, n_images=ni, q_img=q_img, q_lq=q_lq, img=img[:ni].tolist(), img_nq=img_nq[:ni].tolist(), img_q=img_q[:ni]))
    return res


def check_plan(plan, n_tok, cap=8):
    """Invariants of any plan: every image exactly once and in ONE launch; its quarters each in a tile quarter of their own, the table consistent in both
    directions; every tile of a launch but its last one full; launches within the cap.  Returns the number of tiles."""
    S = plan[0]["q_img"].shape[0] if plan else 0
    seen = []
    for L in plan:
        assert 0 <= L["n_slots"] <= S and L["n_images"] <= 4 * S
        used = {}
        for k, img in enumerate(L["img"]):
            seen.append(img)
            nq = (min(max(n_tok[img], 0), 256) + 63) // 64 if n_tok is not None else 4
            assert L["img_nq"][k] == nq
            for y in range(4):
                sq = int(L["img_q"][k][y])
                if y >= nq:
                    assert sq == -1
                    continue
                assert 0 <= sq < 4 * L["n_slots"] and sq not in used
                used[sq] = (img, y)
                assert L["q_img"][sq >> 2][sq & 3] == img and L["q_lq"][sq >> 2][sq & 3] == y
        for s in range(S):
            for w in range(4):
                if 4 * s + w not in used:
                    assert L["q_img"][s][w] == -1 and L["q_lq"][s][w] == -1
        assert sorted(used) == list(range(len(used)))                                 # laid down one after the other: only the last tile can have room
        assert L["n_slots"] == (len(used) + 3) // 4
        tail = cap + cap // 2 if 0 < cap and cap + cap // 2 <= S else S
        assert L["n_slots"] <= max(tail, 1)
    assert sorted(seen) == list(range(len(n_tok) if n_tok is not None else len(seen)))
    return sum(L["n_slots"] for L in plan)


def test_token_packing_plans(hc):
    # four views of <= 64 tokens or two of <= 128 per tile -- and, quarters being independent, ANY mix: every tile but the last is full
    n = [64, 64, 64, 64, 128, 128, 100, 30]
    p = pack(hc, n)
    assert check_plan(p, n) == 3 and len(p) == 1                                      # 11 quarters -> 3 tiles
    assert p[0]["img"] == list(range(8))                                              # the caller's order, nothing sorted
    assert check_plan(pack(hc, [256] * 5), [256] * 5) == 5                                # RGB views: a tile each
    p = pack(hc, [192] * 4)                                                               # 3 quarters each: 12 quarters = 3 full tiles (an image may span two tiles)
    assert check_plan(p, [192] * 4) == 3 and p[0]["img_q"][1].tolist() == [3, 4, 5, -1]
    n = [40, 64, 65, 128, 200]                                                            # the mixed batch of the GPU test: 1 + 1 + 2 + 2 + 4 quarters
    assert check_plan(pack(hc, n), n) == 3
    n = [256, 80, 129, 128, 1, 0, 193, 64, 65]
    p = pack(hc, n)
    assert check_plan(p, n) == 5 and sum(L["n_images"] for L in p) == 9                   # 19 quarters; the image without tokens rides along (no tile)
    assert check_plan(pack(hc, [0, 0, 0]), [0, 0, 0]) == 0                                # nothing to sweep: one launch of post-processing only
    assert pack(hc, [0, 0, 0])[0]["n_slots"] == 0 and pack(hc, [0, 0, 0])[0]["n_images"] == 3
    assert pack(hc, []) == []
    # unknown token counts: one image per tile, launches of 8 with a tail of <= 12 (round 4's behaviour)
    for b in (1, 7, 12, 13, 20, 33):
        p = pack(hc, None, batch=b)
        assert check_plan(p, None) == b
        assert [L["n_slots"] for L in p] == [L["n_images"] for L in p]
        sizes = [L["n_slots"] for L in p]
        assert all(x == 8 for x in sizes[:-1]) and sizes[-1] <= 12
    # 64-token views: 32 of them are 8 tiles = ONE launch (round 4: four launches of 8 images, each streaming the key planes)
    p = pack(hc, [64] * 32)
    assert len(p) == 1 and p[0]["n_slots"] == 8 and p[0]["n_images"] == 32
    # no cap: as many tiles per launch as the table holds
    assert check_plan(pack(hc, [256] * 40, cap=0), [256] * 40, cap=0) == 40


def test_token_packing_random_batches(hc):
    import random
    rnd = random.Random(7)
    for _ in range(300):
        b = rnd.randint(1, 70)
        n = [rnd.choice([0, 1, 40, 63, 64, 65, 100, 127, 128, 129, 176, 192, 193, 255, 256, 300, -3]) for _ in range(b)]
        cap = rnd.choice([8, 8, 8, 4, 1, 0, 21, 30])
        p = pack(hc, n, cap=cap)
        tiles = check_plan(p, n, cap=cap)
        quarters = sum((min(max(v, 0), 256) + 63) // 64 for v in n)
        # only the last tile of a LAUNCH can have room (an image never straddles two launches): at most one tile more per launch than the bound
        assert (quarters + 3) // 4 <= tiles <= (quarters + 3) // 4 + max(len(p) - 1, 0)


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
