"""Mesh cleaning before the DTU evaluation (utils/clean_mesh.py:9-130, evaluation/clean_mesh.py:101-262 of the reference):

  1. clean_mesh_by_mask        keep faces whose three vertices project into more than `min_nb_visible` of the (disk-)dilated
                               object masks
  2. clean_mesh_outside_frustum keep faces that are the FIRST hit of at least one mask-pixel ray of some view, then drop
                               connected components of fewer than 500 faces and unreferenced vertices

trimesh / pyembree / skimage / open3d are not dependencies: the first-hit test runs on the GPU as a z-buffer rasterisation
(csrc/raster.hip via surf_amd.ops.raster_first_hit), dilation uses scipy.ndimage with skimage's disk footprint, components
scipy.sparse.csgraph.  Meshes are (vertices (V,3) float, faces (F,3) int) arrays.
One deviation, on purpose: the reference drops the smallest hit id assuming it is the "no hit" marker -1
(`hull_mask[values[1:]] = 1`, utils/clean_mesh.py:96-97), which would discard a real face when every ray hits; here the
no-hit marker is excluded explicitly."""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import ndimage
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components


def disk(radius):
    """skimage.morphology.disk: (2r+1)^2 footprint of x^2 + y^2 <= r^2."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return (X ** 2 + Y ** 2) <= radius ** 2


def dilate_disk(mask, radius):
    return ndimage.binary_dilation(np.asarray(mask, dtype=bool), structure=disk(radius))


def update_faces(vertices, faces, keep):
    """Trimesh.update_faces(mask) + remove_unreferenced_vertices."""
    faces = np.asarray(faces)[np.asarray(keep, dtype=bool)]
    used = np.zeros(len(vertices), dtype=bool)
    used[faces.reshape(-1)] = True
    remap = np.cumsum(used) - 1
    return np.asarray(vertices)[used], remap[faces]


@torch.no_grad()
def clean_mesh_by_mask(vertices, faces, masks, intrs, c2ws, min_nb_visible=1):
    """utils/clean_mesh.py:9-34.  masks (nv,h,w) bool / float tensors; returns the face keep-mask (F,) bool."""
    points = torch.as_tensor(np.asarray(vertices), dtype=torch.float32).permute(1, 0)
    nv, h, w = masks.shape
    pts_cam = torch.matmul(c2ws.inverse(), torch.cat([points, torch.ones_like(points[:1])], dim=0)[None])[:, :3]
    pts_img = torch.matmul(intrs[:, :3, :3], pts_cam)
    pts_xy = pts_img[:, :2] / torch.clamp(pts_img[:, 2:], 1e-8)
    pts_xy[:, 0] = 2 * pts_xy[:, 0] / (w - 1) - 1
    pts_xy[:, 1] = 2 * pts_xy[:, 1] / (h - 1) - 1
    in_mask = (pts_xy.abs() <= 1).all(dim=1) & (pts_img[:, -1] > 1e-8)
    grid = torch.clamp(pts_xy.permute(0, 2, 1).unsqueeze(1), -10, 10)
    warp_mask = F.grid_sample(masks.unsqueeze(1).float(), grid, align_corners=True).reshape(nv, -1)
    valid = ((warp_mask > 0) * in_mask).sum(dim=0) > min_nb_visible
    return valid[torch.as_tensor(np.asarray(faces), dtype=torch.long)].all(dim=-1).numpy()


def face_components(faces, min_len):
    """trimesh.graph.connected_components(mesh.face_adjacency, min_len): keep-mask of the faces whose edge-connected
    component has at least min_len faces (faces without any neighbour form no component, as in trimesh)."""
    faces = np.asarray(faces)
    F_ = len(faces)
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    owner = np.tile(np.arange(F_), 3)
    order = np.lexsort((e[:, 1], e[:, 0]))
    e, owner = e[order], owner[order]
    same = (e[1:] == e[:-1]).all(axis=1)
    a, b = owner[:-1][same], owner[1:][same]
    if len(a) == 0:
        return np.zeros(F_, dtype=bool)
    n, labels = connected_components(coo_matrix((np.ones(len(a)), (a, b)), shape=(F_, F_)), directed=False)
    size = np.bincount(labels, minlength=n)
    has_nb = np.zeros(F_, dtype=bool)
    has_nb[a] = True
    has_nb[b] = True
    return (size[labels] >= min_len) & has_nb


@torch.no_grad()
def visible_faces(vertices, faces, masks, intrs, c2ws, upscale=4, device="cuda"):
    """Union over the views of the faces that are the first hit of a ray through a masked sample (utils/clean_mesh.py:41-88)."""
    from .. import ops
    v = torch.as_tensor(np.asarray(vertices), dtype=torch.float32, device=device).contiguous()
    f = torch.as_tensor(np.asarray(faces), dtype=torch.int32, device=device).contiguous()
    nv, h, w = masks.shape
    seen = torch.zeros(f.shape[0], dtype=torch.bool, device=device)
    for i in range(nv):
        ids = ops.raster_first_hit(v, f, intrs[i], c2ws[i], (h, w), upscale)
        m = F.interpolate(masks[i].float()[None, None], scale_factor=upscale, mode="nearest")[0, 0].to(device) > 0
        hit = ids[m & (ids >= 0)]
        seen[hit] = True
    return seen.cpu().numpy()


def clean_mesh_outside_frustum(vertices, faces, masks, intrs, c2ws, upscale=4, min_component=500, device="cuda"):
    """utils/clean_mesh.py:37-108."""
    keep = visible_faces(vertices, faces, masks, intrs, c2ws, upscale, device)
    vertices, faces = update_faces(vertices, faces, keep)
    return update_faces(vertices, faces, face_components(faces, min_component))


def clean_mesh(vertices, faces, masks, intrs, c2ws, dilation_radius=11, min_nb_visible=1, upscale=2, min_component=500,
               device="cuda"):
    """utils/clean_mesh.py:110-130 (the entry runner.py:233-234 calls with --clean_mesh)."""
    intrs, c2ws, masks = intrs.cpu(), c2ws.cpu(), masks.cpu()
    if masks.dim() > 3:
        masks = masks.mean(dim=-1)
    dilated = torch.stack([torch.from_numpy(dilate_disk((m > 0.5).numpy(), dilation_radius)) for m in torch.unbind(masks)])
    keep = clean_mesh_by_mask(vertices, faces, dilated, intrs, c2ws, min_nb_visible)
    vertices, faces = np.asarray(vertices), np.asarray(faces)[keep]           # Trimesh.update_faces keeps the vertex list
    return clean_mesh_outside_frustum(vertices, faces, masks, intrs, c2ws, upscale, min_component, device)
