"""DTU / Tanks&Temples readers producing the `ipts` dictionary SuRF.forward consumes.

Mirrors datasets/dtu.py:85-471 (DTUDataset) and datasets/tanks.py (TanksDataset) of the reference: same directory layout,
same conf keys, same keys / shapes / dtypes in the returned dictionary, same order of the random draws (np.random for the
source view and the pseudo points, torch.randint for the training rays).  cv2 and plyfile are replaced by surf_amd.datasets
.mvs_io.  The real datasets are not available in the build container: the readers are tested on synthetic scenes written
in the same file formats (tests/test_datasets.py), so agreement with the reference on real DTU files is NOT pinned."""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from . import mvs_io


def _rays_and_common(out, mode, img_hw, val_res_level, n_rays, imgs, intrs, c2ws, near_fars, masks):
    """Ray / pixel selection shared by the readers (dtu.py:384-433): training draws 3/4 of the rays inside the reference
    mask and 1/4 anywhere; validation takes the strided pixel lattice."""
    H, W = img_hw
    ys, xs = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
    pixel_all = torch.stack([xs, ys], dim=-1)
    if mode == "train":
        assert n_rays > 0, "No sampling rays!"
        p_valid = pixel_all[masks[0] > 0.5]
        pixels_x_i = torch.randint(low=0, high=W, size=[n_rays // 4])
        pixels_y_i = torch.randint(low=0, high=H, size=[n_rays // 4])
        random_idx = torch.randint(low=0, high=p_valid.shape[0], size=[n_rays - n_rays // 4])
        p_select = p_valid[random_idx]
        pixels_x = torch.cat([p_select[:, 0], pixels_x_i], dim=0)
        pixels_y = torch.cat([p_select[:, 1], pixels_y_i], dim=0)
    else:
        out.update({"bound_min": torch.tensor([-1, -1, -1], dtype=torch.float32),
                    "bound_max": torch.tensor([1, 1, 1], dtype=torch.float32)})
        out["hw"] = torch.Tensor([H // val_res_level, W // val_res_level]).int()
        out["masks"] = masks
        tx = torch.linspace(0, W - 1, W // val_res_level)
        ty = torch.linspace(0, H - 1, H // val_res_level)
        pixels_y, pixels_x = torch.meshgrid(ty, tx, indexing="ij")
        pixels_x, pixels_y = pixels_x.reshape(-1), pixels_y.reshape(-1)
    p = torch.stack([pixels_x, pixels_y, torch.ones_like(pixels_y)], dim=-1).float()
    p = torch.matmul(intrs.inverse()[0, None, :3, :3], p[:, :, None]).squeeze(-1)
    rays_d = p / torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    rays_d = torch.matmul(c2ws[0, None, :3, :3], rays_d[:, :, None]).squeeze(-1)
    rays_o = c2ws[0, None, :3, 3].expand(rays_d.shape)
    near, far = near_fars[0].reshape(1, 2).split(split_size=1, dim=1)
    return pixels_x, pixels_y, rays_o, rays_d, near, far


def _normalise_cameras(img_hw, intrs, w2cs, near_fars, factor):
    """dtu.py:336-362: express every camera relative to the reference view, fit the unit sphere (get_scale_mat), re-decompose
    K [R | t] scale_mat and derive near / far = 0.95 (|o| - 1), 1.05 (|o| + 1)."""
    w2c_ref_inv = np.linalg.inv(w2cs[0])
    w2cs = [w2c @ w2c_ref_inv for w2c in w2cs]
    scale_mat, scale_factor = mvs_io.get_scale_mat(img_hw, intrs, w2cs, near_fars, factor=factor)
    c2ws, new_intrs, new_near_fars = [], [], []
    for intr, w2c in zip(intrs, w2cs):
        new_intr, c2w = mvs_io.decompose_projection((intr @ w2c @ scale_mat)[:3, :4])
        c2ws.append(c2w)
        new_intrs.append(new_intr)
        dist = np.sqrt(np.sum(c2w[:3, 3] ** 2)).astype(np.float32)
        new_near_fars.append([0.95 * (dist - 1), 1.05 * (dist + 1)])
    return (torch.from_numpy(np.stack(new_intrs).astype(np.float32)), torch.from_numpy(np.stack(c2ws).astype(np.float32)),
            torch.from_numpy(np.stack(new_near_fars).astype(np.float32)), scale_mat, scale_factor, w2c_ref_inv)


class DTUDataset(Dataset):
    """datasets/dtu.py:85-471.  data_dir layout: Cameras/{vid:08d}_cam.txt (+ pair.txt), Rectified_raw/{scan}/rect_{vid+1:03d}_
    {light}_r5000.png, Depths_raw/{scan}/depth_map_{vid:04d}.pfm + depth_visual_{vid:04d}.png, Pseudo_depths/{scan}/{vid:08d}.pfm,
    Pseudo_points/mvsnet{scan number:03d}_l3.ply."""
    RAW_HW = (1200, 1600)

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
        self.total_views = confs.get_int("total_views", 49)
        self.split = confs.get_string("split", default=None)
        self.scene = confs.get_list("scene", default=None)
        self.light_idx = confs.get_list("light_idx", default=None)
        self.ref_view = confs.get_list("ref_view", default=None)
        self.val_res_level = confs.get_int("val_res_level", default=1) if mode == "val" else 1
        if self.scene is None:
            if self.split is None:
                raise ValueError("There are no scenes!")
            with open(self.split) as f:
                self.scene = [line.rstrip() for line in f.readlines()]
        self.pairs = mvs_io.read_pair_file(os.path.join(self.data_dir, "Cameras/pair.txt"))
        self.metas = self.build_list()

    def build_list(self):
        light_idxs = range(7) if self.light_idx is None else self.light_idx
        all_ref_views = list(range(self.total_views)) if self.ref_view is None else self.ref_view
        return [(scan, int(light), int(ref)) for scan in self.scene for ref in all_ref_views for light in light_idxs]

    def read_cam(self, filename):
        intr, w2c, nf = mvs_io.read_cam_file(filename, self.interval_scale, self.num_interval)
        intr[0] *= self.img_hw[1] / self.RAW_HW[1]
        intr[1] *= self.img_hw[0] / self.RAW_HW[0]
        return intr, w2c, nf

    def read_depth(self, filename):
        return mvs_io.resize_nearest(mvs_io.read_pfm(filename)[0], self.img_hw)

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):
        scan, light_idx, ref_view = self.metas[idx]
        pairs = list(self.pairs[ref_view])
        view_ids = [ref_view] + pairs[:min(self.num_src_view, len(pairs))]
        src_idx = np.random.randint(1, len(view_ids))
        imgs, intrs, w2cs, near_fars, masks = [], [], [], [], []
        for i, vid in enumerate(view_ids):
            suffix = "r7000" if vid > 48 else "r5000"
            img_filename = os.path.join(self.data_dir, "Rectified_raw/{}/rect_{:0>3}_{}_{}.png".format(scan, vid + 1, light_idx, suffix))
            depth_filename = os.path.join(self.data_dir, "Depths_raw/{}/depth_map_{:0>4}.pfm".format(scan, vid))
            pseudo_filename = os.path.join(self.data_dir, "Pseudo_depths/{}/{:0>8}.pfm".format(scan, vid))
            mask_filename = os.path.join(self.data_dir, "Depths_raw/{}/depth_visual_{:0>4}.png".format(scan, vid))
            cam_file = os.path.join(self.data_dir, "Cameras/{:0>8}_cam.txt".format(vid))
            imgs.append(mvs_io.read_image(img_filename, self.img_hw) / 256.0)
            intr, w2c, near_far = self.read_cam(cam_file)
            masks.append((mvs_io.read_image(mask_filename, self.img_hw) > 10).astype(np.float32))
            near_fars.append(near_far)
            intrs.append(intr)
            w2cs.append(w2c)
            if i == 0:
                ref_depth, ref_pseudo = self.read_depth(depth_filename), self.read_depth(pseudo_filename)
            if i == src_idx:
                src_depth, src_pseudo = self.read_depth(depth_filename), self.read_depth(pseudo_filename)
        w2c_ref = w2cs[0]
        intrs_t, c2ws, near_fars_t, scale_mat, scale_factor, w2c_ref_inv = _normalise_cameras(self.img_hw, intrs, w2cs, near_fars,
                                                                                             self.factor)
        ref_depth = torch.from_numpy((ref_depth * scale_factor).astype(np.float32))
        ref_pseudo = torch.from_numpy((ref_pseudo * scale_factor).astype(np.float32))
        src_pseudo = torch.from_numpy((src_pseudo * scale_factor).astype(np.float32))
        src_depth = torch.from_numpy((src_depth * scale_factor).astype(np.float32))
        imgs = torch.from_numpy(np.stack(imgs).astype(np.float32))
        masks = torch.from_numpy(np.stack(masks).astype(np.float32))
        out = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": intrs_t, "c2ws": c2ws,
               "scale_mat": torch.from_numpy(w2c_ref_inv @ scale_mat), "view_ids": torch.from_numpy(np.array(view_ids)).long()}
        if self.mode != "train":
            out["scene"] = scan
            out["file_name"] = scan + "_view" + str(ref_view) + "_light" + str(light_idx)
        pixels_x, pixels_y, rays_o, rays_d, near, far = _rays_and_common(out, self.mode, self.img_hw, self.val_res_level, self.n_rays,
                                                                         imgs, intrs_t, c2ws, near_fars_t, masks)
        yi, xi = pixels_y.long(), pixels_x.long()
        pxyz_ori = mvs_io.read_ply_points(os.path.join(self.data_dir, "Pseudo_points/mvsnet{:0>3}_l3.ply".format(int(scan[4:]))))
        pxyz = pxyz_ori[np.random.randint(low=0, high=pxyz_ori.shape[0], size=[2048])]
        pxyz = np.matmul(w2c_ref, np.concatenate([pxyz, np.ones_like(pxyz[..., :1])], axis=1)[..., None])[:, :3, 0]
        pseudo_pts = torch.from_numpy((pxyz - scale_mat[:3, 3][None]) / scale_mat[0, 0])
        out.update({"pixels_x": pixels_x, "pixels_y": pixels_y, "near_fars": near_fars_t, "rays_o": rays_o, "rays_d": rays_d,
                    "near": near, "far": far, "color": imgs[0][(yi, xi)], "depth": ref_depth[(yi, xi)],
                    "pseudo_depth": ref_pseudo[(yi, xi)], "mask": masks[0][(yi, xi)], "mask_ref": masks[0], "depth_ref": ref_depth,
                    "pseudo_pts": pseudo_pts, "pseudo_depth_ref": ref_pseudo, "pseudo_depth_src": src_pseudo, "src_idx": src_idx,
                    "mask_src": masks[src_idx], "depth_src": src_depth})
        return out


class TanksDataset(Dataset):
    """datasets/tanks.py.  data_dir layout: {scene}/pair.txt, {scene}/images/{vid:08d}.jpg, {scene}/cams/{vid:08d}_cam.txt,
    optional {scene}/masks/{vid:08d}.jpg; raw images are 1080 x 1920."""
    RAW_HW = (1080, 1920)

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
        self.scene = confs.get_list("scene", default=None)
        self.ref_view = confs.get_list("ref_view", default=None)
        self.src_views = confs.get_list("src_views", default=None)
        self.val_res_level = confs.get_int("val_res_level", default=1) if mode == "val" else 1
        if self.scene is None:
            if self.split is None:
                raise ValueError("There are no scenes!")
            with open(self.split) as f:
                self.scene = [line.rstrip() for line in f.readlines()]
        self.metas = self.build_list()

    def build_list(self):
        metas = []
        for scene in self.scene:
            with open(os.path.join(self.data_dir, scene, "pair.txt")) as f:
                lines = [line.rstrip() for line in f.readlines()]
            refs = list(range(int(lines[0]))) if self.ref_view is None else self.ref_view
            for ref_view in refs:
                src = self.src_views if self.src_views is not None else [int(x) for x in lines[2 * int(ref_view) + 2].split()[1::2]]
                metas.append((scene, int(ref_view), [int(v) for v in src]))
        return metas

    def read_cam(self, filename):
        intr, w2c, nf = mvs_io.read_cam_file(filename, self.interval_scale, self.num_interval)
        intr[0] *= self.img_hw[1] / self.RAW_HW[1]
        intr[1] *= self.img_hw[0] / self.RAW_HW[0]
        return intr, w2c, nf

    def __len__(self):
        return len(self.metas)

    def __getitem__(self, idx):
        scan, ref_view, src_views = self.metas[idx]
        view_ids = [ref_view] + src_views[:self.num_src_view]
        imgs, intrs, w2cs, near_fars, depths, masks = [], [], [], [], [], []
        for vid in view_ids:
            img = mvs_io.read_image(os.path.join(self.data_dir, scan, "images", "%08d.jpg" % vid), self.img_hw) / 256.0
            intr, w2c, near_far = self.read_cam(os.path.join(self.data_dir, scan, "cams", "%08d_cam.txt" % vid))
            imgs.append(img)
            intrs.append(intr)
            w2cs.append(w2c)
            near_fars.append(near_far)
            depths.append(np.zeros_like(img[:, :, 0]))
            mask_path = os.path.join(self.data_dir, scan, "masks", "%08d.jpg" % vid)
            masks.append(((mvs_io.read_image(mask_path, self.img_hw) / 255.0) > 0) if os.path.exists(mask_path)
                         else np.ones_like(img[:, :, 0]))
        intrs_t, c2ws, near_fars_t, scale_mat, scale_factor, w2c_ref_inv = _normalise_cameras(self.img_hw, intrs, w2cs, near_fars,
                                                                                             self.factor)
        depths = torch.from_numpy(np.stack([d * scale_factor for d in depths]).astype(np.float32))
        masks = torch.from_numpy(np.stack(masks).astype(np.float32))
        imgs = torch.from_numpy(np.stack(imgs).astype(np.float32))
        out = {"imgs": imgs.permute(0, 3, 1, 2).contiguous(), "intrs": intrs_t, "c2ws": c2ws,
               "scale_mat": torch.from_numpy(w2c_ref_inv @ scale_mat), "view_ids": torch.from_numpy(np.array(view_ids)).long()}
        if self.mode != "train":
            out["scene"] = scan
            out["file_name"] = scan + "_view" + str(ref_view)
        pixels_x, pixels_y, rays_o, rays_d, near, far = _rays_and_common(out, self.mode, self.img_hw, self.val_res_level, self.n_rays,
                                                                         imgs, intrs_t, c2ws, near_fars_t, masks)
        yi, xi = pixels_y.long(), pixels_x.long()
        out.update({"pixels_x": pixels_x, "pixels_y": pixels_y, "near_fars": near_fars_t, "rays_o": rays_o, "rays_d": rays_d,
                    "near": near, "far": far, "color": imgs[0][(yi, xi)], "depth": depths[0][(yi, xi)], "mask": masks[0][(yi, xi)],
                    "masks": masks, "depth_ref": depths[0], "src_idx": 1})
        return out
