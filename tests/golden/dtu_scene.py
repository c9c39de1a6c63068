"""A small synthetic scene written in DTU's (MVSNet pre-processing) file formats, deterministically: the fixture directory of
tests/test_datasets.py, of tests/golden/make_golden_dataset.py (which runs the REFERENCE's reader on it) and of the test that
compares surf_amd's reader with that fixture.  Written with numpy / PIL only."""
import os

import numpy as np
from PIL import Image

K_RAW = np.array([[2892.33, 0, 823.2], [0, 2883.18, 619.07], [0, 0, 1.0]])


def ring_cams(n, radius=600.0):
    """world-to-camera matrices of n cameras on a ring, looking at the origin."""
    cams = []
    for i in range(n):
        a = 0.25 * (i - n // 2)
        o = np.array([radius * np.sin(a), 20.0 * i, -radius * np.cos(a)])
        z = -o / np.linalg.norm(o)
        x = np.cross([0, 1.0, 0], z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        c2w = np.eye(4)
        c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = x, y, z, o
        cams.append(np.linalg.inv(c2w))
    return cams


def write_cam(path, w2c, K, dmin, dint):
    rows = "\n".join(" ".join(f"{v:.8f}" for v in r) for r in w2c)
    krows = "\n".join(" ".join(f"{v:.8f}" for v in r) for r in K)
    with open(path, "w") as f:
        f.write(f"extrinsic\n{rows}\n\nintrinsic\n{krows}\n\n{dmin} {dint}\n")


def write_pfm(filename, image, scale=1.0):
    image = np.flipud(np.asarray(image, dtype="<f4"))
    with open(filename, "wb") as f:
        f.write(("PF\n" if image.ndim == 3 else "Pf\n").encode())
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode())
        f.write(f"{-abs(scale)}\n".encode())
        f.write(image.tobytes())


def write_dtu_scene(root, seed=5, n_views=5, hw=(60, 80)):
    """root/{Cameras, Rectified_raw/scan24, Depths_raw/scan24, Pseudo_depths/scan24, Pseudo_points}: returns (K, w2c list)."""
    g = np.random.default_rng(seed)
    H, W = hw
    for sub in ("Cameras", "Rectified_raw/scan24", "Depths_raw/scan24", "Pseudo_depths/scan24", "Pseudo_points"):
        os.makedirs(os.path.join(root, sub))
    cams = ring_cams(n_views)
    for v, w2c in enumerate(cams):
        write_cam(os.path.join(root, "Cameras", f"{v:08d}_cam.txt"), w2c, K_RAW, 425.0, 2.5)
        Image.fromarray((g.random((H, W, 3)) * 255).astype(np.uint8)).save(
            os.path.join(root, "Rectified_raw/scan24", f"rect_{v + 1:03d}_3_r5000.png"))
        mask = np.zeros((H, W), np.uint8)
        mask[H // 6:H - H // 6, W // 4:W - W // 8] = 255
        Image.fromarray(mask).save(os.path.join(root, "Depths_raw/scan24", f"depth_visual_{v:04d}.png"))
        write_pfm(os.path.join(root, "Depths_raw/scan24", f"depth_map_{v:04d}.pfm"), 500 + 100 * g.random((H, W)).astype(np.float32))
        write_pfm(os.path.join(root, "Pseudo_depths/scan24", f"{v:08d}.pfm"), 500 + 100 * g.random((H, W)).astype(np.float32))
    with open(os.path.join(root, "Cameras", "pair.txt"), "w") as f:
        f.write(f"{n_views}\n" + "".join(f"{r}\n{n_views - 1} " + " ".join(f"{s} 1.0" for s in range(n_views) if s != r) + "\n"
                                        for r in range(n_views)))
    with open(os.path.join(root, "Pseudo_points", "mvsnet024_l3.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex 3000\nproperty float x\nproperty float y\nproperty float z\nend_header\n")
        for p in g.standard_normal((3000, 3)) * 50:
            f.write(" ".join(f"{v:.5f}" for v in p) + "\n")
    return K_RAW, cams


DATASET_CONF = {"dataset_name": "DTUDataset", "scene": ["scan24"], "ref_view": [2], "light_idx": [3], "num_src_view": 2,
                "val_res_level": 4, "factor": 1.0, "interval_scale": 1, "num_interval": 192, "img_hw": [48, 64]}
SEEDS = {"numpy": 11, "torch": 12}


K_TNT = np.array([[1165.7, 0, 962.8], [0, 1166.1, 541.9], [0, 0, 1.0]])
TANKS_CONF = {"dataset_name": "TanksDataset", "scene": ["Family"], "ref_view": [1], "num_src_view": 2, "val_res_level": 2, "factor": 1.0,
              "interval_scale": 1, "num_interval": 192, "img_hw": [54, 96]}


def write_tanks_scene(root, n_views=4, hw=(54, 96), with_masks=(0, 1)):
    """root/Family/{images, cams, masks, pair.txt} in the Tanks&Temples (MVSNet) layout; masks only for the views in with_masks."""
    H, W = hw
    for sub in ("Family/images", "Family/cams", "Family/masks"):
        os.makedirs(os.path.join(root, sub))
    cams = ring_cams(n_views, radius=4.0)
    for v, w2c in enumerate(cams):
        write_cam(os.path.join(root, "Family/cams", f"{v:08d}_cam.txt"), w2c, K_TNT, 1.5, 0.02)
        # PNG bytes under a .jpg name (PIL picks the codec from the extension; lossless keeps the fixture independent of the jpeg encoder)
        Image.fromarray((np.random.default_rng(v).random((H, W, 3)) * 255).astype(np.uint8)).save(
            os.path.join(root, "Family/images", f"{v:08d}.jpg"), format="PNG")
        if v in with_masks:
            m = np.zeros((H, W), np.uint8)
            m[5:40, 10:80] = 255
            Image.fromarray(m).save(os.path.join(root, "Family/masks", f"{v:08d}.jpg"), format="PNG")
    with open(os.path.join(root, "Family", "pair.txt"), "w") as f:
        f.write(f"{n_views}\n" + "".join(f"{r}\n{n_views - 1} " + " ".join(f"{s} 1.0" for s in range(n_views) if s != r) + "\n"
                                        for r in range(n_views)))
    return K_TNT, cams
