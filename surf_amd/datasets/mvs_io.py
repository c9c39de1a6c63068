"""File formats and camera algebra of the reference's dataset readers, without cv2 / plyfile:

  read_cam_file     MVSNet camera text files          datasets/dtu.py:182-203, tanks.py:165-188
  read_pair_file    MVSNet pair.txt                    datasets/dtu.py:130-146, tanks.py:104-123
  read_pfm / write_pfm                                 datasets/dtu.py:38-73
  read_image        PIL + cv2.resize(INTER_NEAREST)    datasets/dtu.py:248-258
  decompose_projection   cv2.decomposeProjectionMatrix as used by load_K_Rt_from_P   datasets/dtu.py:14-35
  get_scale_mat     frustum bounding box -> unit-sphere normalisation                datasets/dtu.py:204-240
  read_ply_points   the x / y / z columns of a PLY vertex element (plyfile in the reference, dtu.py:435-439)
"""
import re

import numpy as np
from PIL import Image


def read_cam_file(filename, interval_scale=1.0, num_interval=192):
    """-> (intrinsics 4x4 float32 with the 3x3 K in the corner, extrinsics 4x4 world-to-camera, [depth_min, depth_max])."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.eye(4, dtype=np.float32)
    intrinsics[:3, :3] = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    depth_min = float(lines[11].split()[0])
    depth_interval = float(lines[11].split()[1]) * interval_scale
    return intrinsics, extrinsics, [depth_min, depth_min + depth_interval * num_interval]


def read_pair_file(filename, max_src=10):
    """-> list over reference views of their source-view ids (best first), dtu.py:136-144."""
    with open(filename) as f:
        n = int(f.readline())
        pairs = [[] for _ in range(n)]
        for _ in range(n):
            ref = int(f.readline().rstrip())
            pairs[ref] = [int(x) for x in f.readline().rstrip().split()[1::2]][:max_src]
    return pairs


def read_pfm(filename):
    """-> (data (H,W) or (H,W,3) float32, bottom-up rows flipped to top-down; scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise ValueError("Not a PFM file.")
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise ValueError("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.frombuffer(f.read(), dtype=endian + "f4")
    shape = (height, width, 3) if header == "PF" else (height, width)
    return np.flipud(data.reshape(shape)).astype(np.float32), abs(scale)


def write_pfm(filename, image, scale=1.0):
    image = np.flipud(np.asarray(image, dtype="<f4"))
    with open(filename, "wb") as f:
        f.write(("PF\n" if image.ndim == 3 else "Pf\n").encode())
        f.write(f"{image.shape[1]} {image.shape[0]}\n".encode())
        f.write(f"{-abs(scale)}\n".encode())
        f.write(image.tobytes())


def resize_nearest(img, hw):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): source index = floor(dst index * src / dst)."""
    H, W = int(hw[0]), int(hw[1])
    h, w = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xs = np.minimum(np.floor(np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def read_image(filename, hw=None):
    img = np.array(Image.open(filename), dtype=np.float32)
    return resize_nearest(img, hw) if hw is not None else img


def decompose_projection(P):
    """K, R, camera centre of a 3x4 projection P = K [R | -R C] (cv2.decomposeProjectionMatrix): RQ decomposition of the
    left 3x3 block with a positive diagonal of K and det(R) = +1.  Returns what load_K_Rt_from_P makes of it
    (datasets/dtu.py:14-35): intrinsics 4x4 (K / K[2,2]) and the camera-to-world pose 4x4 (R^T, C)."""
    P = np.asarray(P, dtype=np.float64)
    M = P[:, :3]
    # RQ through QR of the row-reversed transpose
    J = np.eye(3)[::-1]
    q, r = np.linalg.qr((J @ M).T)
    K = J @ r.T @ J
    R = J @ q.T
    S = np.diag(np.sign(np.diag(K)))          # make diag(K) positive
    K, R = K @ S, S @ R
    if np.linalg.det(R) < 0:                  # P is only defined up to sign
        R = -R
        K = K.copy()                          # K R = -M: the projection is the same up to the homogeneous sign
    C = -np.linalg.solve(M, P[:, 3])
    intrinsics = np.eye(4)
    intrinsics[:3, :3] = K / K[2, 2]
    pose = np.eye(4, dtype=np.float32)
    pose[:3, :3] = R.T
    pose[:3, 3] = C
    return intrinsics, pose


def get_scale_mat(img_hw, intrs, w2cs, near_fars, factor=0.8):
    """Bounding box of the views' frusta between their depth ranges -> (scale_mat, 1 / radius): centre and radius of the
    sphere the scene is normalised into (datasets/dtu.py:204-240)."""
    bnds = np.zeros((3, 2))
    bnds[:, 0], bnds[:, 1] = np.inf, -np.inf
    im_h, im_w = img_hw
    for intr, w2c, (dmin, dmax) in zip(intrs, w2cs, near_fars):
        d = np.array([dmin] * 4 + [dmax] * 4)
        pts = np.stack([(np.array([0, 0, im_w, im_w, 0, 0, im_w, im_w]) - intr[0, 2]) * d / intr[0, 0],
                        (np.array([0, im_h, 0, im_h, 0, im_h, 0, im_h]) - intr[1, 2]) * d / intr[1, 1], d]).astype(np.float32)
        pts = (np.linalg.inv(w2c) @ np.concatenate([pts, np.ones_like(pts[:1])], axis=0))[:3]
        bnds[:, 0] = np.minimum(bnds[:, 0], pts.min(axis=1))
        bnds[:, 1] = np.maximum(bnds[:, 1], pts.max(axis=1))
    center = ((bnds[:, 1] + bnds[:, 0]) / 2).astype(np.float32)
    radius = (bnds[:, 1] - bnds[:, 0]).max() / 2 * factor
    scale_mat = np.diag([radius, radius, radius, 1.0]).astype(np.float32)
    scale_mat[:3, 3] = center
    return scale_mat, 1.0 / radius


_PLY_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
              "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4",
              "float64": "f8"}


def read_ply_points(filename):
    """(N,3) float64 x, y, z of the `vertex` element of an ascii or binary PLY (scalar vertex properties only)."""
    with open(filename, "rb") as f:
        data = f.read()
    end = data.index(b"end_header") + len(b"end_header")
    end = data.index(b"\n", end) + 1
    fmt, props, n, in_vertex = None, [], 0, False
    for l in data[:end].decode("ascii", "ignore").splitlines():
        t = l.split()
        if not t:
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            in_vertex = t[1] == "vertex"
            if in_vertex:
                n = int(t[2])
        elif t[0] == "property" and in_vertex:
            if t[1] == "list":
                raise ValueError("list properties in the vertex element are not supported")
            props.append((t[2], _PLY_TYPES[t[1]]))
    names = [p[0] for p in props]
    if fmt == "ascii":
        rows = np.array([l.split() for l in data[end:].decode("ascii").splitlines()[:n]], dtype=np.float64)
        return np.stack([rows[:, names.index(a)] for a in "xyz"], axis=1)
    endian = "<" if fmt == "binary_little_endian" else ">"
    dt = np.dtype([(nm, endian + ty) for nm, ty in props])
    v = np.frombuffer(data, dtype=dt, count=n, offset=end)
    return np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float64)
