#!/usr/bin/env python3
"""Golden face masks of the REFERENCE's visual-hull mesh filter (row f4): `clean_mesh_by_mask` of /root/reference/utils/clean_mesh.py
(pure torch) is imported and run on a synthetic mesh, three ring cameras and three (already dilated) masks; the face keep-masks it
hands to `mesh.update_faces` are committed as tests/golden/clean_mesh.npz together with the inputs.
Stood in for: the imports `skimage`, `trimesh`, `open3d` of that file (absent from the image, unused by this function) and the mesh
argument - an object with `.vertices` (float64), `.faces` and `.update_faces(mask)`, the three members the function touches."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from surf_amd import synthetic  # noqa: E402
from tests.golden import make_golden as G  # noqa: E402


class Mesh:
    def __init__(self, vertices, faces):
        self.vertices, self.faces, self.kept = np.asarray(vertices, dtype=np.float64), np.asarray(faces), None

    def update_faces(self, mask):
        self.kept = np.asarray(mask).copy()


def inputs():
    g = torch.Generator().manual_seed(5)
    H, W, nv = 60, 80, 3
    intrs, c2ws, _ = synthetic.ring_cameras(nv, H, W)
    verts = (torch.rand(4000, 3, generator=g) * 2 - 1) * 0.45
    faces = torch.randint(0, 4000, (9000, 3), generator=g)
    masks = torch.zeros(nv, H, W, dtype=torch.bool)
    masks[0, 4:56, 6:74] = True
    masks[1, 2:50, 20:80] = True
    masks[2, 10:60, 0:60] = True
    masks[2, 26:36, 30:42] = False                        # a hole
    return verts.numpy().astype(np.float64), faces.numpy(), masks, intrs, c2ws


def main():
    for name in ("skimage", "skimage.morphology", "trimesh", "open3d"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["skimage"].morphology = sys.modules["skimage.morphology"]
    sys.path.insert(0, G.REF)
    from utils.clean_mesh import clean_mesh_by_mask
    v, f, masks, intrs, c2ws = inputs()
    out = {"vertices": torch.from_numpy(v), "faces": torch.from_numpy(f), "masks": masks, "intrs": intrs, "c2ws": c2ws}
    for m in (0, 1, 2):
        mesh = Mesh(v, f)
        clean_mesh_by_mask(mesh, masks, intrs, c2ws, min_nb_visible=m)
        out[f"keep{m}"] = torch.from_numpy(mesh.kept.astype(np.uint8))
        print("min_nb_visible", m, "faces kept", int(mesh.kept.sum()), "of", len(f))
    G.ONLY.clear()
    G.npz("clean_mesh.npz", **out)


if __name__ == "__main__":
    main()
