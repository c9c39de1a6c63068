#!/usr/bin/env python3
"""Golden GRADIENTS of the reference's own training loss through its own render (row f2): generated in the build container by
importing the reference's modules from /root/reference (same shims as make_golden.py, whose fixtures this script READS and
does not rewrite).  Only data is committed.

  train_grads.npz
    the reference's ImplicitSurface.forward("train", ...) on the golden scene (perturb = 0, cos_anneal_ratio = 0.7, step = 3,
    with the dataset's `pseudo_pts`), its Loss.forward (losses/loss.py:27-111: colour, eikonal, sparse, smooth, mfc, depth,
    pseudo-depth, pseudo-SDF terms, confs/surf.conf weights with depth_weight raised), then loss.backward():
      loss            the scalar
      grad/<name>     d loss / d every parameter of the implicit surface (state_dict names)
      grad_vol<lvl>   d loss / d the sparse feature rows, fine -> coarse
      target_*, pseudo_pts   the seeded inputs
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import make_golden as G  # noqa: E402
from tests.golden.make_golden_train import LOSS_CONF, load  # noqa: E402

COS_ANNEAL, STEP, SEED = 0.7, 3, 0


def main():
    FeatureNetwork, Volume, MatchingField, ImplicitSurface, projector = G.import_reference()
    from models.losses.loss import Loss
    conf = G.Conf(G.MODEL_CONF)
    scene, weights, fpn, pipe = load("scene.npz"), load("weights.npz"), load("fpn.npz"), load("pipeline.npz")
    isurf = ImplicitSurface(conf["implicit_surface"]).train()
    isurf.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")})
    feats_r = [fpn[f"out{i}"] for i in range(4)][::-1]
    vols_r, tabs_r, masks_r = [], [], []
    for s in range(4):
        table = pipe[f"s{s}_table"].long()
        vols_r.append(pipe[f"s{s}_reg_out"][:, 1:].contiguous())
        tabs_r.append(table)
        masks_r.append((table >= 0).float()[None, None])
    vols_r, tabs_r, masks_r = vols_r[::-1], tabs_r[::-1], masks_r[::-1]
    vols_r = [v.clone().requires_grad_(True) for v in vols_r]
    mvol = pipe["s3_mvol"][None, None]
    R = scene["rays_o"].shape[0]
    g = torch.Generator().manual_seed(77)
    pseudo = (torch.rand(160, 3, generator=g) * 2 - 1) * 0.75
    targets = {"color": torch.rand(R, 3, generator=g), "mask": (torch.rand(R, generator=g) > 0.2).float(),
               "pseudo_depth": torch.rand(R, generator=g) * (torch.rand(R, generator=g) > 0.3).float() * 3.0,
               "depth": torch.rand(R, generator=g) * (torch.rand(R, generator=g) > 0.5).float() * 3.0}
    ipts = dict(scene)
    ipts["pseudo_pts"] = pseudo
    torch.manual_seed(SEED)
    outs = isurf("train", ipts, mvol, vols_r, tabs_r, masks_r, feats_r, feats_r, COS_ANNEAL, STEP)
    lo = Loss(G.Conf(LOSS_CONF))(outs, targets, STEP, "val")
    lo["loss"].backward()
    out = {"loss": lo["loss"].detach().reshape(1), "pseudo_pts": pseudo, "pseudo_sdf": outs["pseudo_sdf"].detach(),
           "color_fine": outs["color_fine"].detach(), "smooth_error": outs["smooth_error"].detach().reshape(1)}
    for k, v in lo.items():
        out["loss_out_" + k] = torch.as_tensor(v, dtype=torch.float32).detach().reshape(-1)
    for k, v in targets.items():
        out["target_" + k] = v
    for name, p in isurf.named_parameters():
        out["grad/" + name] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach()
    for lvl, v in enumerate(vols_r):
        out[f"grad_vol{lvl}"] = v.grad.detach()
    G.ONLY.clear()
    G.npz("train_grads.npz", **out)
    print("loss", float(lo["loss"]), {k: float(torch.as_tensor(v)) for k, v in lo.items() if k != "loss"})


if __name__ == "__main__":
    main()
