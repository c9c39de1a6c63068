"""DTU / Tanks&Temples readers producing the `ipts` dictionary SuRF.forward consumes.

What is contractual - and pinned to the reference's readers (datasets/dtu.py:85-471 `DTUDataset`, datasets/tanks.py
`TanksDataset`) - is the OUTPUT: the dictionary's keys, shapes and dtypes, the conf keys, the datasets' own directory
layouts and file names, and the order of the random draws (numpy's global generator for the source view and the pseudo
points, torch's for the training pixels), so that a seeded run of either reader yields the same batch.  The code that
produces it is organised our way:

    DTUFiles       where the files of one view of a scan live on disk (the dataset's layout)
    RawViews       images / cameras / masks of the selected views as read
    normalise_rig  the reference's camera normalisation: poses relative to the reference view, unit-sphere fit,
                   re-decomposition into K, camera-to-world, near / far
    choose_pixels  training draw (3/4 inside the reference mask, 1/4 anywhere) or the strided validation lattice
    pixel_rays     rays of the reference camera through those pixels
    scene_block / ray_block   the per-scene and per-ray dictionary entries both readers share

cv2 and plyfile are replaced by surf_amd.datasets.mvs_io.  The real datasets are not available in the build container: the
readers are tested on synthetic scenes written in the same file formats (tests/test_datasets.py), so agreement with the
reference on real DTU files is NOT pinned.
"""
import os
from collections import namedtuple

import numpy as np
import torch
from torch.utils.data import Dataset

from . import mvs_io

RawViews = namedtuple("RawViews", "imgs intrs w2cs near_fars masks")
Rig = namedtuple("Rig", "intrs c2ws near_fars scale_mat scale_factor ref_c2w_raw")


def _f32(x):
    return torch.from_numpy(np.asarray(x, dtype=np.float32))


def scale_intrinsics(intr, img_hw, raw_hw):
    """Intrinsics of the raw image size -> the working size (rows 0 / 1 scale with width / height)."""
    intr = intr.copy()
    intr[0] *= img_hw[1] / raw_hw[1]
    intr[1] *= img_hw[0] / raw_hw[0]
    return intr


def normalise_rig(img_hw, raw, factor):
    """datasets/dtu.py:336-362.  Every camera is expressed in the reference view's frame, the union of the frusta (between
    the depth planes) is fitted into the unit sphere (mvs_io.get_scale_mat) and each K [R | t] scale_mat is decomposed again
    (mvs_io.decompose_projection); near / far bracket the unit sphere as seen from each camera centre: 0.95 (|o| - 1),
    1.05 (|o| + 1)."""
    ref_c2w_raw = np.linalg.inv(raw.w2cs[0])
    relative = [w2c @ ref_c2w_raw for w2c in raw.w2cs]
    scale_mat, scale_factor = mvs_io.get_scale_mat(img_hw, raw.intrs, relative, raw.near_fars, factor=factor)
    intrs, c2ws, spans = [], [], []
    for K, w2c in zip(raw.intrs, relative):
        K_n, c2w = mvs_io.decompose_projection((K @ w2c @ scale_mat)[:3, :4])
        centre_dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
        intrs.append(K_n)
        c2ws.append(c2w)
        spans.append([0.95 * (centre_dist - 1), 1.05 * (centre_dist + 1)])
    return Rig(_f32(np.stack(intrs)), _f32(np.stack(c2ws)), _f32(np.stack(spans)), scale_mat, scale_factor, ref_c2w_raw)


def choose_pixels(mode, img_hw, n_rays, ref_mask, res_level):
    """(x, y) float pixel coordinates of the batch.  train (dtu.py:389-404): three torch.randint draws in the reference's
    order - x then y of the n // 4 unconstrained pixels, then n - n // 4 indices into the row-major list of pixels with
    mask > 0.5; the masked pixels come first in the batch.  Otherwise (dtu.py:406-417): the `res_level`-strided lattice over
    the whole image, row-major."""
    H, W = img_hw
    if mode != "train":
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H // res_level), torch.linspace(0, W - 1, W // res_level), indexing="ij")
        return gx.reshape(-1), gy.reshape(-1)
    if n_rays <= 0:
        raise AssertionError("No sampling rays!")
    n_free = n_rays // 4
    free_x = torch.randint(low=0, high=W, size=[n_free])
    free_y = torch.randint(low=0, high=H, size=[n_free])
    inside_yx = torch.nonzero(ref_mask > 0.5)                                 # row-major, like boolean indexing of an (H, W, 2) grid
    pick = inside_yx[torch.randint(low=0, high=inside_yx.shape[0], size=[n_rays - n_free])].float()
    return torch.cat([pick[:, 1], free_x.float()]), torch.cat([pick[:, 0], free_y.float()])


def pixel_rays(px, py, K_ref, c2w_ref):
    """Unit ray directions (world frame) and origins of the reference camera through pixels (px, py) (dtu.py:423-427)."""
    homog = torch.stack([px, py, torch.ones_like(py)], dim=-1).float()
    cam = homog @ torch.inverse(K_ref)[:3, :3].T
    cam = cam / torch.linalg.norm(cam, ord=2, dim=-1, keepdim=True)
    dirs = cam @ c2w_ref[:3, :3].T
    return c2w_ref[:3, 3].expand(dirs.shape), dirs


def ray_block(mode, img_hw, n_rays, res_level, rig, ref_mask):
    """The per-ray entries every reader shares."""
    px, py = choose_pixels(mode, img_hw, n_rays, ref_mask, res_level)
    rays_o, rays_d = pixel_rays(px, py, rig.intrs[0], rig.c2ws[0])
    near, far = rig.near_fars[0, 0].reshape(1, 1), rig.near_fars[0, 1].reshape(1, 1)
    return {"pixels_x": px, "pixels_y": py, "rays_o": rays_o, "rays_d": rays_d, "near": near, "far": far,
            "near_fars": rig.near_fars}, (py.long(), px.long())


def scene_block(mode, img_hw, res_level, raw_imgs, rig, view_ids, masks, scan, file_name):
    """The per-scene entries every reader shares (+ the validation-only ones, dtu.py:406-411)."""
    imgs = _f32(np.stack(raw_imgs))
    out = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": rig.intrs, "c2ws": rig.c2ws,
           "scale_mat": torch.from_numpy(rig.ref_c2w_raw @ rig.scale_mat), "view_ids": torch.from_numpy(np.array(view_ids)).long()}
    if mode != "train":
        out.update(bound_min=torch.tensor([-1, -1, -1], dtype=torch.float32), bound_max=torch.tensor([1, 1, 1], dtype=torch.float32),
                   scene=scan, file_name=file_name, masks=masks,
                   hw=torch.Tensor([img_hw[0] // res_level, img_hw[1] // res_level]).int())
    return out, imgs


def _scene_list(confs):
    scene = confs.get_list("scene", default=None)
    if scene is not None:
        return scene
    split = confs.get_string("split", default=None)
    if split is None:
        raise ValueError("There are no scenes!")
    with open(split) as f:
        return [line.rstrip() for line in f.readlines()]


class _MVSDataset(Dataset):
    """Conf keys and camera reading common to both readers."""
    RAW_HW = None

    def __init__(self, confs, mode):
        super().__init__()
        self.mode = mode
        self.data_dir = confs["data_dir"]
        self.num_src_view = confs.get_int("num_src_view")
        self.interval_scale = confs.get_float("interval_scale")
        self.num_interval = confs.get_int("num_interval")
        self.img_hw = [int(v) for v in confs["img_hw"]]
        self.n_rays = confs.get_int("n_rays", 0)
        self.factor = confs.get_float("factor")
        self.split = confs.get_string("split", default=None)
        self.ref_view = confs.get_list("ref_view", default=None)
        self.val_res_level = confs.get_int("val_res_level", default=1) if mode == "val" else 1
        self.scene = _scene_list(confs)

    def read_cam(self, filename):
        intr, w2c, near_far = mvs_io.read_cam_file(filename, self.interval_scale, self.num_interval)
        return scale_intrinsics(intr, self.img_hw, self.RAW_HW), w2c, near_far

    def __len__(self):
        return len(self.metas)


class DTUFiles:
    """DTU (MVSNet pre-processing) layout under data_dir: Cameras/{vid:08d}_cam.txt + Cameras/pair.txt,
    Rectified_raw/{scan}/rect_{vid+1:03d}_{light}_r5000.png (r7000 for the extra views > 48), Depths_raw/{scan}/
    depth_map_{vid:04d}.pfm + depth_visual_{vid:04d}.png, Pseudo_depths/{scan}/{vid:08d}.pfm,
    Pseudo_points/mvsnet{scan number:03d}_l3.ply."""

    def __init__(self, root):
        self.root = root

    def path(self, *parts):
        return os.path.join(self.root, *parts)

    def image(self, scan, vid, light):
        return self.path("Rectified_raw", scan, "rect_{:0>3}_{}_{}.png".format(vid + 1, light, "r7000" if vid > 48 else "r5000"))

    def camera(self, vid):
        return self.path("Cameras", "{:0>8}_cam.txt".format(vid))

    def depth(self, scan, vid):
        return self.path("Depths_raw", scan, "depth_map_{:0>4}.pfm".format(vid))

    def mask(self, scan, vid):
        return self.path("Depths_raw", scan, "depth_visual_{:0>4}.png".format(vid))

    def pseudo_depth(self, scan, vid):
        return self.path("Pseudo_depths", scan, "{:0>8}.pfm".format(vid))

    def pseudo_points(self, scan):
        return self.path("Pseudo_points", "mvsnet{:0>3}_l3.ply".format(int(scan[4:])))


class DTUDataset(_MVSDataset):
    """datasets/dtu.py:85-471."""
    RAW_HW = (1200, 1600)

    def __init__(self, confs, mode):
        super().__init__(confs, mode)
        self.total_views = confs.get_int("total_views", 49)
        self.light_idx = confs.get_list("light_idx", default=None)
        self.files = DTUFiles(self.data_dir)
        self.pairs = mvs_io.read_pair_file(self.files.path("Cameras", "pair.txt"))
        self.metas = self.build_list()

    def build_list(self):
        """One item per (scan, reference view, light condition), lights innermost (dtu.py:157-180)."""
        lights = range(7) if self.light_idx is None else self.light_idx
        refs = range(self.total_views) if self.ref_view is None else self.ref_view
        return [(scan, int(light), int(ref)) for scan in self.scene for ref in refs for light in lights]

    def read_depth(self, filename):
        return mvs_io.resize_nearest(mvs_io.read_pfm(filename)[0], self.img_hw)

    def _depth_pair(self, scan, vid, scale):
        """(ground-truth depth, pseudo (MVS) depth) of a view in the normalised frame."""
        return tuple(_f32(self.read_depth(f(scan, vid)) * scale) for f in (self.files.depth, self.files.pseudo_depth))

    def __getitem__(self, idx):
        scan, light, ref_view = self.metas[idx]
        view_ids = [ref_view] + list(self.pairs[ref_view])[:self.num_src_view]
        src_idx = np.random.randint(1, len(view_ids))                              # draw 1 (numpy): the supervised source view
        raw = RawViews([], [], [], [], [])
        for vid in view_ids:
            K, w2c, near_far = self.read_cam(self.files.camera(vid))
            raw.imgs.append(mvs_io.read_image(self.files.image(scan, vid, light), self.img_hw) / 256.0)
            raw.masks.append((mvs_io.read_image(self.files.mask(scan, vid), self.img_hw) > 10).astype(np.float32))
            raw.intrs.append(K)
            raw.w2cs.append(w2c)
            raw.near_fars.append(near_far)
        rig = normalise_rig(self.img_hw, raw, self.factor)
        masks = _f32(np.stack(raw.masks))
        out, imgs = scene_block(self.mode, self.img_hw, self.val_res_level, raw.imgs, rig, view_ids, masks, scan,
                                f"{scan}_view{ref_view}_light{light}")
        rays, at = ray_block(self.mode, self.img_hw, self.n_rays, self.val_res_level, rig, masks[0])    # draws 2-4 (torch)
        out.update(rays)
        depth_ref, pseudo_ref = self._depth_pair(scan, view_ids[0], rig.scale_factor)
        depth_src, pseudo_src = self._depth_pair(scan, view_ids[src_idx], rig.scale_factor)
        # pseudo surface points (dtu.py:435-446): 2048 of the MVS point cloud, into the reference camera's frame, then the
        # unit-sphere normalisation
        cloud = mvs_io.read_ply_points(self.files.pseudo_points(scan))
        cloud = cloud[np.random.randint(low=0, high=cloud.shape[0], size=[2048])]   # draw 5 (numpy)
        cloud_h = np.concatenate([cloud, np.ones_like(cloud[..., :1])], axis=1)
        in_ref = np.matmul(raw.w2cs[0], cloud_h[..., None])[:, :3, 0]
        out["pseudo_pts"] = torch.from_numpy((in_ref - rig.scale_mat[:3, 3][None]) / rig.scale_mat[0, 0])
        out.update(color=imgs[0][at], depth=depth_ref[at], pseudo_depth=pseudo_ref[at], mask=masks[0][at], mask_ref=masks[0],
                   depth_ref=depth_ref, pseudo_depth_ref=pseudo_ref, pseudo_depth_src=pseudo_src, src_idx=src_idx,
                   mask_src=masks[src_idx], depth_src=depth_src)
        return out


class TanksDataset(_MVSDataset):
    """datasets/tanks.py.  Layout under data_dir: {scene}/pair.txt, {scene}/images/{vid:08d}.jpg, {scene}/cams/{vid:08d}_cam.txt,
    optional {scene}/masks/{vid:08d}.jpg; raw images are 1080 x 1920.  No ground-truth depths: zeros."""
    RAW_HW = (1080, 1920)

    def __init__(self, confs, mode):
        super().__init__(confs, mode)
        self.src_views = confs.get_list("src_views", default=None)
        self.metas = self.build_list()

    def build_list(self):
        metas = []
        for scene in self.scene:
            with open(os.path.join(self.data_dir, scene, "pair.txt")) as f:
                lines = [line.rstrip() for line in f.readlines()]
            for ref in (range(int(lines[0])) if self.ref_view is None else self.ref_view):
                src = self.src_views if self.src_views is not None else lines[2 * int(ref) + 2].split()[1::2]
                metas.append((scene, int(ref), [int(v) for v in src]))
        return metas

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.num_src_view]
        raw = RawViews([], [], [], [], [])
        for vid in view_ids:
            name = "%08d" % vid
            img = mvs_io.read_image(os.path.join(self.data_dir, scan, "images", name + ".jpg"), self.img_hw) / 256.0
            K, w2c, near_far = self.read_cam(os.path.join(self.data_dir, scan, "cams", name + "_cam.txt"))
            mask_path = os.path.join(self.data_dir, scan, "masks", name + ".jpg")
            has_mask = os.path.exists(mask_path)
            raw.masks.append(((mvs_io.read_image(mask_path, self.img_hw) / 255.0) > 0) if has_mask else np.ones_like(img[:, :, 0]))
            raw.imgs.append(img)
            raw.intrs.append(K)
            raw.w2cs.append(w2c)
            raw.near_fars.append(near_far)
        rig = normalise_rig(self.img_hw, raw, self.factor)
        masks = _f32(np.stack(raw.masks))
        out, imgs = scene_block(self.mode, self.img_hw, self.val_res_level, raw.imgs, rig, view_ids, masks, scan, f"{scan}_view{ref_view}")
        rays, at = ray_block(self.mode, self.img_hw, self.n_rays, self.val_res_level, rig, masks[0])
        out.update(rays)
        no_depth = torch.zeros(self.img_hw[0], self.img_hw[1], dtype=torch.float32)
        out.update(color=imgs[0][at], depth=no_depth[at], mask=masks[0][at], masks=masks, depth_ref=no_depth, src_idx=1)
        return out
