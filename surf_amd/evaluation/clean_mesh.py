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


def _mask_coverage(mask, px, py):
    """Bilinear sample (align_corners=True pixel coordinates, zeros outside) of a non-negative (h, w) mask at float pixel
    positions: > 0 exactly where one of the four neighbouring texels with a non-zero weight is set."""
    h, w = mask.shape
    x0, y0 = torch.floor(px), torch.floor(py)
    fx, fy = px - x0, py - y0
    total = torch.zeros_like(px)
    for dy, wy in ((0, 1.0 - fy), (1, fy)):
        for dx, wx in ((0, 1.0 - fx), (1, fx)):
            xi, yi = x0.long() + dx, y0.long() + dy
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            total += torch.where(ok, mask[yi.clamp(0, h - 1), xi.clamp(0, w - 1)] * wx * wy, torch.zeros_like(px))
    return total


@torch.no_grad()
def clean_mesh_by_mask(vertices, faces, masks, intrs, c2ws, min_nb_visible=1):
    """The visual-hull test of utils/clean_mesh.py:9-34: a vertex counts as seen by a view when it lies in front of the camera,
    projects inside the image and onto a set texel of that view's (dilated) mask (bilinear footprint); a face survives when
    each of its vertices is seen by MORE than `min_nb_visible` views.  masks (nv,h,w) bool / float tensors; returns the face
    keep-mask (F,) bool (the caller applies it: Trimesh.update_faces)."""
    xyz1 = torch.cat([torch.as_tensor(np.asarray(vertices), dtype=torch.float32),
                      torch.ones(len(vertices), 1, dtype=torch.float32)], dim=1)                   # (V, 4)
    n_seen = torch.zeros(len(vertices), dtype=torch.long)
    for mask, K, c2w in zip(masks.float(), intrs, c2ws):
        h, w = mask.shape
        uvw = (xyz1 @ torch.inverse(c2w).T)[:, :3] @ K[:3, :3].T                                    # homogeneous pixel coordinates
        depth = uvw[:, 2]
        px, py = uvw[:, 0] / depth.clamp(min=1e-8), uvw[:, 1] / depth.clamp(min=1e-8)
        inside = (px >= 0) & (px <= w - 1) & (py >= 0) & (py <= h - 1) & (depth > 1e-8)
        far = 10.0 * max(h, w)                         # keep the index arithmetic finite for points far outside the image
        on_mask = _mask_coverage(mask, px.clamp(-far, far), py.clamp(-far, far)) > 0
        n_seen += (inside & on_mask).long()
    vertex_ok = n_seen > min_nb_visible
    return vertex_ok[torch.as_tensor(np.asarray(faces), dtype=torch.long)].all(dim=-1).numpy()


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
