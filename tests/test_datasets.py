"""Row f3: dataset readers.  Pinned against the reference's own readers: tests/golden/dataset_items.npz holds what
datasets/dtu.py's and datasets/tanks.py's `__getitem__` returned for synthetic scenes written in the datasets' file formats
(tests/golden/make_golden_dataset.py; cv2.resize, cv2.decomposeProjectionMatrix and plyfile stood in for).  Also checked on their
own: the file-format handling (MVSNet camera / pair files, PFM, PNG / JPG, PLY), the camera algebra (K, R, C recovered from a
projection; the unit-sphere normalisation re-projects every point onto the same pixel) and the `ipts` contract."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

from surf_amd import conf
from surf_amd.datasets import DTUDataset, TanksDataset, get_loader, mvs_io


def _rot(a, b, c):
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    return (np.array([[cc, -sc, 0], [sc, cc, 0], [0, 0, 1]]) @ np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
            @ np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]]))


def test_decompose_projection_recovers_k_r_c():
    g = np.random.default_rng(0)
    for _ in range(20):
        K = np.array([[800 + 400 * g.random(), 2 * g.standard_normal(), 300 + 100 * g.random()],
                      [0, 700 + 400 * g.random(), 200 + 100 * g.random()], [0, 0, 1.0]])
        R = _rot(*g.uniform(-1.5, 1.5, 3))
        C = g.standard_normal(3) * 3
        s = g.uniform(0.2, 5.0)                                 # a projection is only defined up to scale
        P = s * K @ np.concatenate([R, (-R @ C)[:, None]], axis=1)
        intr, pose = mvs_io.decompose_projection(P)
        assert np.allclose(intr[:3, :3], K, rtol=1e-9, atol=1e-7)
        assert np.allclose(pose[:3, :3], R.T, atol=1e-6) and np.allclose(pose[:3, 3], C, atol=1e-5)
        assert intr.shape == (4, 4) and pose.dtype == np.float32


def test_pfm_pair_cam_and_ply_files(tmp_path):
    d = np.random.default_rng(1).random((7, 9)).astype(np.float32)
    mvs_io.write_pfm(tmp_path / "d.pfm", d)
    back, scale = mvs_io.read_pfm(tmp_path / "d.pfm")
    assert np.array_equal(back, d) and scale == 1.0
    c = np.random.default_rng(2).random((5, 4, 3)).astype(np.float32)
    mvs_io.write_pfm(tmp_path / "c.pfm", c)
    assert np.array_equal(mvs_io.read_pfm(tmp_path / "c.pfm")[0], c)
    (tmp_path / "pair.txt").write_text("3\n0\n3 1 0.9 2 0.5 0 0.1\n1\n2 0 0.7 2 0.6\n2\n2 1 0.8 0 0.2\n")
    assert mvs_io.read_pair_file(tmp_path / "pair.txt") == [[1, 2, 0], [0, 2], [1, 0]]
    (tmp_path / "cam.txt").write_text("extrinsic\n1 0 0 1\n0 1 0 2\n0 0 1 3\n0 0 0 1\n\nintrinsic\n2892.3 0 823.2\n0 2883.2 619.1\n0 0 1\n\n"
                                      "425.0 2.5\n")
    intr, w2c, nf = mvs_io.read_cam_file(tmp_path / "cam.txt", interval_scale=1.0, num_interval=192)
    assert intr[0, 0] == np.float32(2892.3) and w2c[2, 3] == 3 and nf == [425.0, 425.0 + 2.5 * 192]
    img = np.arange(6 * 8, dtype=np.float32).reshape(6, 8)
    r = mvs_io.resize_nearest(img, (3, 4))
    assert np.array_equal(r, img[::2, ::2])                     # cv2.INTER_NEAREST: floor(dst * src / dst)
    pts = np.random.default_rng(3).standard_normal((11, 3)).astype(np.float32)
    with open(tmp_path / "p.ply", "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nend_header\n")
        rec = np.zeros(11, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1")])
        rec["x"], rec["y"], rec["z"] = pts.T
        f.write(rec.tobytes())
    assert np.allclose(mvs_io.read_ply_points(tmp_path / "p.ply"), pts)
    with open(tmp_path / "a.ply", "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 2\nproperty float x\nproperty float y\nproperty float z\nend_header\n"
                "1 2 3\n4 5 6\n")
    assert np.array_equal(mvs_io.read_ply_points(tmp_path / "a.ply"), [[1, 2, 3], [4, 5, 6]])


from tests.golden.dtu_scene import ring_cams as _ring_cams  # noqa: E402
from tests.golden.dtu_scene import write_dtu_scene


def _write_cam(path, w2c, K, dmin, dint):
    from tests.golden.dtu_scene import write_cam
    write_cam(str(path), w2c, K, dmin, dint)


@pytest.fixture()
def dtu_dir(tmp_path):
    root = tmp_path / "dtu"
    K, cams = write_dtu_scene(str(root))
    return root, K, cams


def _dtu_conf(root, **kw):
    c = {"dataset_name": "DTUDataset", "data_dir": str(root), "scene": ["scan24"], "ref_view": [2], "light_idx": [3],
         "num_src_view": 2, "val_res_level": 4, "factor": 1.0, "interval_scale": 1, "num_interval": 192, "img_hw": [48, 64],
         "total_views": 5}
    c.update(kw)
    return conf.from_dict(c)


def test_dtu_val_item_is_the_ipts_contract(dtu_dir):
    root, K, cams = dtu_dir
    ds = DTUDataset(_dtu_conf(root), "val")
    assert len(ds) == 1
    np.random.seed(0)
    it = ds[0]
    want = {"imgs", "intrs", "c2ws", "scale_mat", "view_ids", "bound_min", "bound_max", "scene", "file_name", "hw", "masks",
            "pixels_x", "pixels_y", "near_fars", "rays_o", "rays_d", "near", "far", "color", "depth", "pseudo_depth", "mask",
            "mask_ref", "depth_ref", "pseudo_pts", "pseudo_depth_ref", "pseudo_depth_src", "src_idx", "mask_src", "depth_src"}
    assert set(it) == want
    assert it["view_ids"].tolist() == [2, 0, 1] and it["file_name"] == "scan24_view2_light3" and it["hw"].tolist() == [12, 16]
    assert tuple(it["imgs"].shape) == (3, 3, 48, 64) and float(it["imgs"].max()) < 1.0
    R = 12 * 16
    assert tuple(it["rays_o"].shape) == (R, 3) and tuple(it["color"].shape) == (R, 3) and tuple(it["pseudo_pts"].shape) == (2048, 3)
    assert torch.allclose(it["rays_d"].norm(dim=1), torch.ones(R), atol=1e-5)
    assert torch.allclose(it["rays_o"], it["c2ws"][0, :3, 3].expand(R, 3))
    # the reference view sits at the identity of the relative frame before normalisation: its rotation stays the identity
    assert torch.allclose(it["c2ws"][0, :3, :3], torch.eye(3), atol=1e-5)
    dist = it["c2ws"][:, :3, 3].norm(dim=1)
    assert torch.allclose(it["near_fars"], torch.stack([0.95 * (dist - 1), 1.05 * (dist + 1)], 1), atol=1e-5)
    assert float(it["near"]) == float(it["near_fars"][0, 0]) and tuple(it["near"].shape) == (1, 1)
    # re-projection: a point of the normalised frame lands on the same pixel through (intrs, c2ws) as through the original
    # camera files after scale_mat (datasets/dtu.py:344-352, runner.py:236)
    g = torch.Generator().manual_seed(0)
    x = (torch.rand(50, 3, generator=g) * 2 - 1) * 0.5
    xw = (it["scale_mat"].double() @ torch.cat([x.double(), torch.ones(50, 1, dtype=torch.float64)], 1).T).T      # original world
    Ks = K.copy()
    Ks[0] *= 64 / 1600
    Ks[1] *= 48 / 1200
    for j, vid in enumerate(it["view_ids"].tolist()):
        cam = (torch.from_numpy(cams[vid]) @ xw.T).T[:, :3]
        uv0 = (torch.from_numpy(Ks) @ cam.T).T
        uv0 = uv0[:, :2] / uv0[:, 2:]
        cn = (torch.inverse(it["c2ws"][j].double()) @ torch.cat([x.double(), torch.ones(50, 1, dtype=torch.float64)], 1).T).T[:, :3]
        uv1 = (it["intrs"][j, :3, :3].double() @ cn.T).T
        uv1 = uv1[:, :2] / uv1[:, 2:]
        assert float((uv0 - uv1).abs().max()) < 2e-2, j       # fp32 camera files / matrices
        assert float(cn[:, 2].min()) > 0
    # the whole scene (frusta between the depth planes) fits the unit sphere: camera centres are outside it
    assert float(dist.min()) > 1.0


def test_dtu_train_item_draws_rays_inside_the_mask(dtu_dir):
    root, _, _ = dtu_dir
    ds = DTUDataset(_dtu_conf(root, n_rays=512), "train")
    torch.manual_seed(0)
    np.random.seed(0)
    it = ds[0]
    assert it["rays_o"].shape[0] == 512 and "file_name" not in it and 1 <= it["src_idx"] <= 2
    inside = it["mask_ref"][it["pixels_y"].long(), it["pixels_x"].long()]
    assert float(inside[:384].min()) == 1.0 and torch.equal(inside, it["mask"])
    loader, sampler, dataset = get_loader(_dtu_conf(root, n_rays=64), "train", False, num_workers=0)
    batch = next(iter(loader))
    assert batch["rays_d"].shape == (64, 3) and dataset is not None and sampler is not None


def test_tanks_item(tmp_path):
    root = tmp_path / "tnt"
    for sub in ("Family/images", "Family/cams"):
        os.makedirs(root / sub)
    K = np.array([[1165.7, 0, 962.8], [0, 1166.1, 541.9], [0, 0, 1.0]])
    for v, w2c in enumerate(_ring_cams(4, radius=4.0)):
        _write_cam(root / "Family/cams" / f"{v:08d}_cam.txt", w2c, K, 1.5, 0.02)
        Image.fromarray((np.random.default_rng(v).random((54, 96, 3)) * 255).astype(np.uint8)).save(root / "Family/images" / f"{v:08d}.jpg")
    (root / "Family" / "pair.txt").write_text("4\n" + "".join(
        f"{r}\n3 " + " ".join(f"{s} 1.0" for s in range(4) if s != r) + "\n" for r in range(4)))
    c = conf.from_dict({"dataset_name": "TanksDataset", "data_dir": str(root), "scene": ["Family"], "ref_view": [1], "num_src_view": 2,
                        "val_res_level": 2, "factor": 1.0, "interval_scale": 1, "num_interval": 192, "img_hw": [54, 96]})
    ds = TanksDataset(c, "val")
    it = ds[0]
    assert it["view_ids"].tolist() == [1, 0, 2] and it["file_name"] == "Family_view1" and it["src_idx"] == 1
    assert tuple(it["imgs"].shape) == (3, 3, 54, 96) and tuple(it["rays_o"].shape) == (27 * 48, 3)
    assert float(it["masks"].min()) == 1.0 and float(it["depth_ref"].abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        get_loader(conf.from_dict({"dataset_name": "BMVSDataset"}), "val", False)


def _compare_with_reference_item(item, gold, tag):
    want_keys = {k.split("/")[1] for k in gold if k.startswith(tag + "/") and not k.startswith(tag + "/str/")}
    want_strs = {k.split("/")[2]: k.split("/")[3] for k in gold if k.startswith(tag + "/str/")}
    assert set(item) == want_keys | set(want_strs), set(item) ^ (want_keys | set(want_strs))
    for k, v in want_strs.items():
        assert item[k] == v
    for k in sorted(want_keys):
        ref, got = gold[f"{tag}/{k}"], item[k]
        if not torch.is_tensor(got):
            assert int(got) == int(ref), k
            continue
        assert got.dtype == ref.dtype and tuple(got.shape) == tuple(ref.shape), (k, got.dtype, ref.dtype, got.shape, ref.shape)
        if got.dtype.is_floating_point:
            assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6 * float(ref.abs().max() + 1)), (k, float((got - ref).abs().max()))
        else:
            assert torch.equal(got, ref), k


@pytest.mark.parametrize("mode", ["val", "train"])
def test_dtu_reader_equals_the_reference_reader(tmp_path, mode):
    """Row f3 against the REFERENCE itself: tests/golden/dataset_items.npz holds what datasets/dtu.py's own
    `DTUDataset.__getitem__` returned for the synthetic scene of tests/golden/dtu_scene.py (generated in the build container by
    tests/golden/make_golden_dataset.py; cv2.resize / cv2.decomposeProjectionMatrix / plyfile stood in for, everything else the
    reference's code).  surf_amd's reader, same files, same seeds: every tensor of the dictionary - cameras after the unit-sphere
    normalisation, near / far, the seeded pixel draws, rays, colours / depths at the pixels, pseudo points - equals it, keys
    and dtypes included."""
    from tests.conftest import load_npz
    from tests.golden.dtu_scene import DATASET_CONF, SEEDS
    gold = load_npz("dataset_items.npz")
    root = tmp_path / "dtu"
    write_dtu_scene(str(root))
    extra = {} if mode == "val" else {"n_rays": 96}
    ds = DTUDataset(conf.from_dict(dict(DATASET_CONF, data_dir=str(root), **extra)), mode)
    np.random.seed(SEEDS["numpy"])
    torch.manual_seed(SEEDS["torch"])
    _compare_with_reference_item(ds[0], gold, mode)


@pytest.mark.parametrize("mode", ["val", "train"])
def test_tanks_reader_equals_the_reference_reader(tmp_path, mode):
    """The same for datasets/tanks.py's `TanksDataset.__getitem__` (BASELINE configs[4]'s reader): views with and without a mask
    file, zero depths, the fixed src_idx."""
    from tests.conftest import load_npz
    from tests.golden.dtu_scene import SEEDS, TANKS_CONF, write_tanks_scene
    gold = load_npz("dataset_items.npz")
    root = tmp_path / "tnt"
    write_tanks_scene(str(root))
    extra = {} if mode == "val" else {"n_rays": 64}
    ds = TanksDataset(conf.from_dict(dict(TANKS_CONF, data_dir=str(root), **extra)), mode)
    np.random.seed(SEEDS["numpy"])
    torch.manual_seed(SEEDS["torch"])
    _compare_with_reference_item(ds[0], gold, "tanks_" + mode)
