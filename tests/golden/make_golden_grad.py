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
    # ---- the smooth (H.1) term alone (weight 1e-4 in the loss above: its share of those gradients is below their tolerance)
    for p_ in isurf.parameters():
        p_.grad = None
    vols_s = [v.detach().clone().requires_grad_(True) for v in vols_r]
    torch.manual_seed(SEED)
    outs_s = isurf("train", ipts, mvol, vols_s, tabs_r, masks_r, feats_r, feats_r, COS_ANNEAL, STEP)
    outs_s["smooth_error"].backward()
    for name, p_ in isurf.named_parameters():
        out["smooth_grad/" + name] = (p_.grad if p_.grad is not None else torch.zeros_like(p_)).detach()
    for lvl, v in enumerate(vols_s):
        out[f"smooth_grad_vol{lvl}"] = (v.grad if v.grad is not None else torch.zeros_like(v)).detach()
    G.ONLY.clear()
    G.npz("train_grads.npz", **out)
    print("loss", float(lo["loss"].detach()))
    volume_side(FeatureNetwork, Volume, MatchingField, conf, scene, weights, fpn, pipe)


def volume_side(FeatureNetwork, Volume, MatchingField, conf, scene, weights, fpn, pipe):
    """volume_grads.npz: the reference's autograd through its own volume-build modules, one piece at a time (the sparse U-Net
    needs torchsparse and cannot be run): FPN parameters, the cost volume (feature maps + agg_mlp), sparse2dense, the matching
    field's depth maps (train-mode jitter, views 0 / src_idx), compute_ptloss - for seeded upstream gradients, which are stored."""
    import numpy as np
    from models.losses.photometric_loss import compute_ptloss
    out = {}
    intrs, c2ws = scene["intrs"], scene["c2ws"]
    g = torch.Generator().manual_seed(2024)
    # ---- FPN (feature_network.py:158-178)
    fnet = FeatureNetwork(conf["feature_network"]).train()
    fnet.load_state_dict({k[len("feature_network."):]: v for k, v in weights.items() if k.startswith("feature_network.")})
    outs = fnet(scene["imgs"])                                       # coarse -> fine
    ups = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * u).sum() for o, u in zip(outs, ups)).backward()
    for i, u in enumerate(ups):
        out[f"fpn_up{i}"] = u
    for name, p in fnet.named_parameters():
        out["fpn_grad/" + name] = p.grad.detach()
    # ---- cost volume (volume.py:54-97), stages 0 and 2
    vol = Volume(conf["volume"]).train()
    vol.load_state_dict({k[len("volume."):]: v for k, v in weights.items() if k.startswith("volume.")})
    base = np.array(conf["volume"].get_list("base_volume_dim"))
    for stage in (0, 2):
        vol.zero_grad()
        vol.volume_dim = base * 2 ** stage
        vol.voxel_size = (vol.bounding[:, 1] - vol.bounding[:, 0]) / (vol.volume_dim - 1)
        feats = [fpn[f"out{i}"].clone().requires_grad_(True) for i in range(4)]
        coords = pipe[f"s{stage}_coords"].float()
        cv, _ = vol.back_proj_multiscale(feats, coords, intrs, c2ws, stage)
        up = torch.randn(cv.shape, generator=g)
        (cv * up).sum().backward()
        out[f"cv{stage}_up"] = up
        for i, f in enumerate(feats):
            out[f"cv{stage}_gfeat{i}"] = f.grad.detach() if f.grad is not None else torch.zeros_like(f)
        for name, p in vol.named_parameters():
            out[f"cv{stage}_grad/" + name] = p.grad.detach().clone()
    # ---- sparse2dense (volume.py:99-121), stage 1 on top of stage 0's matching volume
    vol.volume_dim = base * 2
    vol.voxel_size = (vol.bounding[:, 1] - vol.bounding[:, 0]) / (vol.volume_dim - 1)
    logit = pipe["s1_reg_out"][:, :1].clone().requires_grad_(True)
    prev = pipe["s0_mvol"][None, None].clone().requires_grad_(True)
    dense, _ = vol.sparse2dense(logit, pipe["s1_coords"].float(), prev)
    up = torch.randn(dense.shape, generator=g)
    (dense * up).sum().backward()
    out.update(s2d_up=up[0, 0], s2d_glogit=logit.grad[:, 0].detach(), s2d_gprev=prev.grad[0, 0].detach())
    # ---- matching field (matching_field.py:73-141), stage 1, train-mode jitter, views 0 and src_idx = 2
    mf = MatchingField(conf["matching_field"]).train()
    ipts = dict(scene)
    ipts["src_idx"] = 2
    ratios = list(G.MODEL_CONF["range_ratios"])
    mv = pipe["s1_mvol"][None, None].clone().requires_grad_(True)
    torch.manual_seed(31)
    depths, _ = mf(ipts, mv, 1, ratios, list(pipe["s0_depths"]), perturb=True)
    H, W = scene["imgs"].shape[-2:]
    up0, up2 = torch.randn(H, W, generator=g), torch.randn(H, W, generator=g)
    ((depths[0] * up0).sum() + (depths[2] * up2).sum()).backward()
    out.update(mf_up0=up0, mf_up2=up2, mf_gmvol=mv.grad[0, 0].detach())
    # ---- photometric term (losses/photometric_loss.py:54-125) w.r.t. the depth map
    mask = (torch.rand(H, W, generator=g) > 0.15).float()
    dep = pipe["s3_depths"][0].clone().requires_grad_(True)
    compute_ptloss(dep, scene["imgs"], mask, intrs, c2ws).backward()
    out.update(pt_mask=mask, pt_gdepth=dep.grad.detach())
    G.npz("volume_grads.npz", **out)


if __name__ == "__main__":
    main()
