"""The rest of pose_estimation/isocell.py (random modes, grouping helpers) against golden g13 (generated from the reference on CPU)."""
import importlib
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden", "g13_isocell_rest.npz")
iso = importlib.import_module("6dgs_amd.isocell")


@pytest.fixture(scope="module")
def g13():
    return np.load(G)


def test_random_modes_raise_where_the_reference_raises_and_match_it_for_one_ring(g13):
    for mode, tgt, n0, raised in g13["raises"].tolist():
        torch.manual_seed(1300 + 10 * mode + n0)
        if raised:
            with pytest.raises(RuntimeError):
                iso.isocell_distribution(tgt, torch.float32, "cpu", N0=n0, isrand=mode)
        else:
            d = iso.isocell_distribution(tgt, torch.float32, "cpu", N0=n0, isrand=mode)
            want = g13[f"rand_m{mode}_t{tgt}_n{n0}"]
            assert tuple(d.shape) == want.shape                       # modes 1-4: [1, 3 N0]: x.., y.., z.. (the reference's column_stack of [1,N0] rows);
            assert np.array_equal(d.numpy(), want), (mode, tgt, n0)   # any other mode (0, 7: the else-branch): [N0, 3].  Same generator, same draw order: bit-exact on CPU
    assert {int(m) for m in g13["raises"][:, 0]} == {0, 1, 2, 3, 4, 7}


def test_random_mode_consumes_the_generator_like_the_reference_before_raising():
    torch.manual_seed(5)
    with pytest.raises(RuntimeError):
        iso.isocell_distribution(64, torch.float32, "cpu", isrand=1)
    after = torch.rand(1)
    torch.manual_seed(5)
    torch.rand(1)
    assert torch.equal(after, torch.rand(1))


@pytest.mark.parametrize("tgt,n0", [(27, 3), (48, 3), (64, 1)])
def test_grouping_helpers_match_the_reference(g13, tgt, n0):
    p = f"grp_{tgt}_{n0}_"
    dirs = torch.from_numpy(g13[p + "dirs"])
    grp, ring, cell = iso.group_by_360_isocell(dirs, tgt, N0=n0)
    assert np.array_equal(grp.numpy(), g13[p + "group"])
    assert np.array_equal(ring.numpy(), g13[p + "ring"])
    assert np.array_equal(cell.numpy(), g13[p + "cell"])
    groups = iso.get_dirs_group_idx(dirs, tgt, N0=n0)
    assert [int(g.shape[0]) for g in groups] == g13[p + "sizes"].tolist()
    assert np.array_equal(torch.cat(groups).numpy(), g13[p + "members"])
    assert all(g.dtype == torch.int64 for g in groups)
    # the quirk that is kept: the directions of the largest key are in no group
    assert int(g13[p + "members"].shape[0]) < dirs.shape[0]


def test_grouping_of_nothing_and_of_a_single_key():
    one = torch.tensor([[0.1, 0.0, 0.995]], dtype=torch.float32)
    one = one / one.norm()
    assert iso.get_dirs_group_idx(one, 27, N0=3) == []                # its key is the largest key


@pytest.mark.gpu
def test_module_surface_on_the_gpu_is_the_hip_path():
    ops = importlib.import_module("6dgs_amd.ops")
    d = iso.isocell_distribution(64, torch.float32, "cuda", N0=1)
    assert d.is_cuda and torch.equal(d, ops.isocell_distribution(64, 1, device="cuda"))
    n = torch.nn.functional.normalize(torch.randn(7, 3, device="cuda"), dim=-1)
    assert torch.equal(iso.rotate_isocell(d, n), ops.rotate_isocell(d, n))
    with pytest.raises(RuntimeError):
        iso.isocell_distribution(64, torch.float64, "cuda")
    # random modes on the GPU draw from the DEVICE generator, as the reference does (isocell.py:24,47-60 pass device=device): reproducible
    # under a CUDA seed, on the unit hemisphere, and not the CPU stream's numbers (the golden pins the CPU generator only)
    g = np.load(G)
    torch.manual_seed(1311)
    r = iso.isocell_distribution(3, torch.float32, "cuda", N0=3, isrand=1)
    torch.manual_seed(1311)
    r2 = iso.isocell_distribution(3, torch.float32, "cuda", N0=3, isrand=1)
    assert r.is_cuda and tuple(r.shape) == (1, 9) and torch.equal(r, r2)
    xyz = r.reshape(3, 3)
    assert torch.allclose((xyz * xyz).sum(0), torch.ones(3, device="cuda"), atol=1e-6) and bool((xyz[2] >= 0).all())
    c = iso.isocell_distribution(3, torch.float32, "cuda", N0=3, isrand=7)          # the else-branch: centred cells, [N0, 3]
    assert tuple(c.shape) == (3, 3) and torch.allclose(c[:, :2].norm(dim=1), torch.full((3,), 0.5, device="cuda"), atol=1e-6)
    dirs = torch.from_numpy(g["grp_27_3_dirs"]).cuda()
    grp, ring, cell = iso.group_by_360_isocell(dirs, 27, N0=3)
    assert grp.is_cuda and np.array_equal(grp.cpu().numpy(), g["grp_27_3_group"])
    assert ring.shape == cell.shape == grp.shape and ring.dtype == torch.int64   # (ids next to a cell edge may differ with the device's libm: pinned on CPU)
