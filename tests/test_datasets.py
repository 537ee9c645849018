"""Scene ingestion (6dgs_amd/datasets.py) against the reference's own loaders: golden g12 holds what scene/dataset_loader.load_data
(COLMAP binary + text, NSVF / Tanks&Temples, Blender) and pose_estimation/file_utils.parse_exp_dir return for the on-disk fixtures
of synthetic.write_dataset_fixtures (oracle/gen_golden.py:g12 ran the reference on them in the build container).  The fixtures are
regenerated here by the same deterministic writer.  cfg_args parsing: the reference needs antlr4 (absent here, so it could not be
run: parity unpinned); the parser is held to the grammar (cfg_grammar/Namespace.g4) and to the example string of the reference's
own cfg_grammar/parse_config.py:51."""
import importlib
import os

import numpy as np
import pytest

ds = importlib.import_module("6dgs_amd.datasets")


@pytest.fixture(scope="module")
def fixtures(tmp_path_factory, syn):
    root = tmp_path_factory.mktemp("datasets")
    return str(root), syn.write_dataset_fixtures(str(root), 0)


@pytest.mark.parametrize("name", ["colmap_bin", "colmap_txt", "tt", "blender"])
@pytest.mark.parametrize("ev,wb", [(True, False), (False, True)])
def test_camera_loaders_match_the_reference(fixtures, golden, name, ev, wb):
    g = golden("g12_datasets")
    src = fixtures[1][name]
    info = ds.load_data(ds.dotdict(source_path=src, images=None, eval=ev, white_background=wb))
    tag = f"{name}_e{int(ev)}w{int(wb)}"
    for split, cams in (("train", info.train_cameras), ("test", info.test_cameras)):
        assert len(cams) == int(g[f"{tag}_{split}_n"]), (tag, split)
        if not cams:
            continue
        ref = g[f"{tag}_{split}_RT"]
        got = np.stack([np.concatenate([np.asarray(c.R, np.float64).reshape(9), np.asarray(c.T, np.float64), [c.FovY, c.FovX, c.width, c.height, c.uid]])
                        for c in cams])
        assert np.abs(got - ref).max() < 1e-12, (tag, split)                       # same float64 arithmetic: exact up to library rounding
        assert [c.image_name for c in cams] == [str(x) for x in g[f"{tag}_{split}_names"]]
        imgs = np.stack([np.array(c.image) for c in cams])
        assert imgs.dtype == np.uint8 and np.array_equal(imgs, g[f"{tag}_{split}_img"]), (tag, split)     # incl. the RGBA composite's bytes
    assert abs(info.nerf_normalization["radius"] - float(g[f"{tag}_radius"])) < 1e-6 * float(g[f"{tag}_radius"])       # fp32 path in the reference
    assert np.abs(np.asarray(info.nerf_normalization["translate"]) - g[f"{tag}_translate"]).max() < 1e-6
    assert ds.get_dataset_prefix(src) == str(g[f"{tag}_prefix"])


def test_the_splits_are_what_the_reference_produces(fixtures):
    """llffhold = 8 on name-sorted COLMAP cameras; Tanks&Temples split 0 / 2 (else 1); Blender train + test merged without eval."""
    src = fixtures[1]
    info = ds.load_data(ds.dotdict(source_path=src["colmap_bin"], images=None, eval=True, white_background=False))
    names = sorted(c.image_name for c in info.train_cameras + info.test_cameras)
    assert [c.image_name for c in info.test_cameras] == names[::8] and len(info.train_cameras) == 8
    info = ds.load_data(ds.dotdict(source_path=src["tt"], images=None, eval=True, white_background=False))
    assert all(c.image_name.startswith("0_") for c in info.train_cameras) and all(c.image_name.startswith("1_") for c in info.test_cameras)
    # R is the world-to-camera rotation TRANSPOSED and T its translation: [R^T | T] inverts to the file's camera-to-world pose
    c = info.train_cameras[0]
    w2c = np.eye(4)
    w2c[:3, :3], w2c[:3, 3] = np.asarray(c.R).T, c.T
    assert np.abs(np.linalg.inv(w2c) - np.loadtxt(os.path.join(src["tt"], "pose", c.image_name.replace(".png", ".txt")))).max() < 1e-9


def test_experiment_discovery_matches_the_reference(tmp_path, golden):
    g = golden("g12_datasets")
    exp = str(tmp_path / "output")
    for d, its in (("mip_360_garden_ab12", (7000, 30000)), ("mip_360_room_cd34", (30000,)), ("tt_Ignatius_ef56", (100, 20)), ("tt_empty_gh78", ()),
                   ("synthetic_lego_0001", (5,))):
        os.makedirs(os.path.join(exp, d, "point_cloud"), exist_ok=True)
        for it in its:
            os.makedirs(os.path.join(exp, d, "point_cloud", f"iteration_{it}"), exist_ok=True)
            open(os.path.join(exp, d, "point_cloud", f"iteration_{it}", "point_cloud.ply"), "wb").close()
    os.makedirs(os.path.join(exp, "tt_Ignatius_ef56", "point_cloud", "iteration_900"), exist_ok=True)
    os.makedirs(os.path.join(exp, "mip_360_room_cd34", "point_cloud", "notes_1"), exist_ok=True)
    for prefix in ("", "mip_360_", "tt_", "synthetic_"):
        found = ds.parse_exp_dir(exp, prefix)
        got = [f"{k}|{v['category_name']}|{os.path.relpath(v['checkpoint_filepath'], exp)}" for k, v in found.items()]
        assert got == [str(x) for x in g[f"exp_{prefix or 'all'}"]], prefix
    assert ds.parse_exp_dir(exp, "tt_")["ef56"]["checkpoint_filepath"].endswith(os.path.join("iteration_100", "point_cloud.ply"))


def test_cfg_args_parser_follows_the_grammar(tmp_path):
    text = ("Namespace(sh_degree=3, source_path='/home/mbortolon/data/datasets/360_v2/bicycle', model_path='./output/ec0d365d-5', images='images', "
            "resolution=-1, white_background=False, data_device='cuda', eval=True)")          # cfg_grammar/parse_config.py:51
    d = ds.parse_cfg_args(text)
    assert d == {"sh_degree": 3, "source_path": "/home/mbortolon/data/datasets/360_v2/bicycle", "model_path": "./output/ec0d365d-5", "images": "images",
                 "resolution": -1, "white_background": True, "data_device": "cuda", "eval": True}      # bool("False") is True in the reference
    assert ds.parse_cfg_args(text, reference_bools=False)["white_background"] is False
    d = ds.parse_cfg_args('Namespace( a = +12 ,b=-0.5,\n c=3. , d="x y", e=true )')
    assert d == {"a": 12, "b": -0.5, "c": 3.0, "d": "x y", "e": True} and isinstance(d["c"], float)
    for bad in ("Namespace(a=None)", "Namespace(a=[1])", "Namespace(a=1e5)", "Namespace(a=1 b=2)", "Namespace(=1)", "a=1", "Namespace(a='x\\'y')"):
        with pytest.raises(ValueError):
            ds.parse_cfg_args(bad)
    (tmp_path / "cfg_args").write_text(text)
    args = ds.get_checkpoint_arguments(str(tmp_path))
    assert args.sh_degree == 3 and args.fps_sampling is None and args.source_path.endswith("bicycle")       # missing keys read as None
