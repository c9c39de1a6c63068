#!/usr/bin/env python3
"""Golden vectors of the training-only outputs of the hot path (SURVEY rows a15, a11 smooth term), generated in the build
container by importing the reference's own modules from /root/reference (same shims as make_golden.py, whose fixtures
this script READS and does not rewrite).  Only data is committed.

  train_outputs.npz
    unit level  surface_patch_warp2 (projector.py:560-645) on given surface points / gradients / the stacked feature maps
                (implicit_surface.py:231-235: FPN levels 0, 1, 2, the coarser two F.interpolate'd to full resolution)
    chain       ImplicitSurface.render (perturb = 0, cos_anneal_ratio = 1): ref_gray_val, sampled_gray_val, smooth_error
    losses      compute_LNCC2 (losses/ncc.py:7-51) on the chain's patches and Loss.forward (losses/loss.py:27-111, mode "val":
                every term but the per-stage photometric ones) on the chain's outputs with seeded targets
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import make_golden as G  # noqa: E402


def load(name):
    z = np.load(os.path.join(HERE, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


LOSS_CONF = {"color_weight": 1.0, "sparse_weight": 0.02, "igr_weight": 0.1, "sparse_scale_factor": 100, "mfc_weight": 1.0,
             "smooth_weight": 0.0001, "tv_weight": 0.0, "depth_weight": 0.5, "ptloss_weight": 1.0, "pseudo_auxi_depth_weight": 1.0,
             "pseudo_sdf_weight": 1.0, "stage_weights": [0.25, 0.5, 0.75, 1.0], "pseudo_depth_weight": 1.0}   # confs/surf.conf:49-63 (depth_weight raised from 0 so that the term is exercised)


def main():
    FeatureNetwork, Volume, MatchingField, ImplicitSurface, projector = G.import_reference()
    conf = G.Conf(G.MODEL_CONF)
    scene, weights, fpn, pipe = load("scene.npz"), load("weights.npz"), load("fpn.npz"), load("pipeline.npz")
    isurf = ImplicitSurface(conf["implicit_surface"]).eval()
    isurf.load_state_dict({k[len("implicit_surface."):]: v for k, v in weights.items() if k.startswith("implicit_surface.")})
    feats = [fpn[f"out{i}"] for i in range(4)]                     # coarse -> fine
    feats_r = feats[::-1]
    vols_r, tabs_r, masks_r = [], [], []
    for s in range(4):
        table = pipe[f"s{s}_table"].long()
        vols_r.append(pipe[f"s{s}_reg_out"][:, 1:].contiguous())
        tabs_r.append(table)
        masks_r.append((table >= 0).float()[None, None])
    vols_r, tabs_r, masks_r = vols_r[::-1], tabs_r[::-1], masks_r[::-1]
    mvol = pipe["s3_mvol"][None, None]
    intrs, c2ws = scene["intrs"], scene["c2ws"]
    out = {}
    # ---- unit level: points near the r = 0.5 surface seen by all cameras, arbitrary (non-unit) gradients
    g = torch.Generator().manual_seed(11)
    n = 40
    dirs = F.normalize(torch.randn(n, 3, generator=g) * torch.tensor([0.6, 0.6, 1.0]) + torch.tensor([0.0, 0.0, -1.2]), dim=1)
    pts = dirs * (0.5 + 0.05 * torch.randn(n, 1, generator=g))
    grads = dirs * (0.5 + torch.rand(n, 1, generator=g)) + 0.1 * torch.randn(n, 3, generator=g)
    grads[0] = 0.0                                                  # the |g| <= 0 -> 1e-8 branch (implicit_surface.py:226)
    with torch.no_grad():
        w0 = feats_r[0]
        w1 = F.interpolate(feats_r[1], size=w0.shape[-2:], mode="bilinear")
        w2 = F.interpolate(feats_r[2], size=w0.shape[-2:], mode="bilinear")
        warp_feats = torch.cat([w0, w1, w2], dim=1)
        gn = torch.linalg.norm(grads.reshape(n, 1, 3), ord=2, dim=-1, keepdim=True)
        gn = torch.where(gn <= 0, torch.ones_like(gn) * 1e-8, gn)
        g_cam = grads.reshape(n, 1, 3) / gn
        g_cam = torch.matmul(c2ws[0, :3, :3].permute(1, 0).contiguous()[None, ...], g_cam.permute(0, 2, 1).contiguous())
        g_cam = g_cam.permute(0, 2, 1).contiguous()
        ref, src = projector.surface_patch_warp2(pts.reshape(n, 1, 3), g_cam, warp_feats, intrs, c2ws)
    out.update(unit_pts=pts, unit_grads=grads, unit_warp_feats=warp_feats, unit_ref=ref, unit_src=src)
    # ---- chain: the reference's own render
    R = scene["rays_o"].shape[0]
    near, far = scene["near"].repeat(R, 1), scene["far"].repeat(R, 1)
    torch.manual_seed(0)
    outs = isurf.render(scene["rays_o"], scene["rays_d"], near, far, mvol, vols_r, tabs_r, masks_r, scene["imgs"], feats_r,
                        feats_r, intrs, c2ws, 1.0, None)
    out.update(ref_gray_val=outs["ref_gray_val"].detach(), sampled_gray_val=outs["sampled_gray_val"].detach(),
               smooth_error=outs["smooth_error"].detach(), mid_inside_sphere=outs["mid_inside_sphere"].detach(),
               sdf_depth=outs["sdf_depth"].detach())
    # ---- train-mode matching field (perturb=True: views 0 and src_idx jittered, matching_field.py:33-35,129-133), stage 1
    mf = MatchingField(conf["matching_field"]).eval()
    ipts = dict(scene)
    ipts["src_idx"] = 2
    ratios = list(G.MODEL_CONF["range_ratios"]) if "range_ratios" in G.MODEL_CONF else [1.0, 0.4, 0.1, 0.01]
    pre = list(pipe["s0_depths"])
    torch.manual_seed(31)
    with torch.no_grad():
        d1, _ = mf(ipts, pipe["s1_mvol"][None, None], 1, ratios, pre, perturb=True)
        torch.manual_seed(31)
        d0, _ = mf(ipts, pipe["s0_mvol"][None, None], 0, ratios, None, perturb=True)
    out.update(mf_perturb_s1=torch.stack(d1), mf_perturb_s0=torch.stack(d0), mf_perturb_src_idx=torch.tensor(2))
    # ---- per-stage photometric term (losses/photometric_loss.py:54-125) on the pipeline's finest depth maps
    from models.losses.photometric_loss import compute_ptloss
    g = torch.Generator().manual_seed(41)
    H, W = scene["imgs"].shape[-2:]
    mask_ref = (torch.rand(H, W, generator=g) > 0.15).float()
    mask_src = (torch.rand(H, W, generator=g) > 0.15).float()
    with torch.no_grad():
        pt_ref = compute_ptloss(pipe["s3_depths"][0], scene["imgs"], mask_ref, intrs, c2ws)
        pt_src = compute_ptloss(pipe["s3_depths"][2], scene["imgs"], mask_src, intrs, c2ws, ref_idx=2, topk=1)
        pt_far = compute_ptloss(pipe["s3_depths"][0] * 3.0, scene["imgs"], mask_ref, intrs, c2ws)      # many pixels leave the frusta
    out.update(pt_mask_ref=mask_ref, pt_mask_src=mask_src, pt_ref=pt_ref.reshape(1), pt_src=pt_src.reshape(1), pt_far=pt_far.reshape(1))
    # ---- losses on the chain's outputs
    from models.losses.loss import Loss
    from models.losses.ncc import compute_LNCC2
    out["ncc"] = compute_LNCC2(outs["ref_gray_val"].detach(), outs["sampled_gray_val"].detach())
    out["unit_ncc"] = compute_LNCC2(out["unit_ref"], out["unit_src"])
    g = torch.Generator().manual_seed(23)
    targets = {"color": torch.rand(R, 3, generator=g), "mask": (torch.rand(R, generator=g) > 0.2).float(),
               "pseudo_depth": torch.rand(R, generator=g) * (torch.rand(R, generator=g) > 0.3).float() * 3.0,
               "depth": torch.rand(R, generator=g) * (torch.rand(R, generator=g) > 0.5).float() * 3.0}
    loss_fn = Loss(G.Conf(LOSS_CONF))
    preds = {k: (v.detach() if torch.is_tensor(v) else v) for k, v in outs.items()}
    preds["pseudo_sdf"] = torch.randn(64, 1, generator=g) * 0.05
    lo = loss_fn(preds, targets, step=1, mode="val")
    for k, v in targets.items():
        out["loss_target_" + k] = v
    out["loss_pred_pseudo_sdf"] = preds["pseudo_sdf"]
    for k in ("color_fine", "valid_mask", "gradient_error", "sparse_sdf", "render_depth"):
        out["loss_pred_" + k] = preds[k].float() if preds[k].dtype == torch.bool else preds[k]
    for k, v in lo.items():
        out["loss_out_" + k] = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
    # ... and mode "train": adds the per-stage photometric / auxiliary depth terms (targets from the pipeline's depth maps)
    targets_t = dict(targets, imgs=scene["imgs"], intrs=intrs, c2ws=c2ws, src_idx=2, mask_ref=mask_ref, mask_src=mask_src,
                     pseudo_depth_ref=pipe["s3_depths"][0] * (torch.rand(H, W, generator=g) > 0.4).float(),
                     pseudo_depth_src=pipe["s3_depths"][2] * (torch.rand(H, W, generator=g) > 0.4).float(),
                     depth_ref=pipe["s3_depths"][0] * 1.02, depth_src=pipe["s3_depths"][2] * 0.98)
    preds_t = dict(preds)
    for i in range(4):
        preds_t[f"depth_stage{i}"] = pipe[f"s{i}_depths"][0]
        preds_t[f"depth_src_stage{i}"] = pipe[f"s{i}_depths"][2]
    with torch.no_grad():
        lt = loss_fn(preds_t, targets_t, step=3, mode="train")
    for k in ("pseudo_depth_ref", "pseudo_depth_src", "depth_ref", "depth_src"):
        out["loss_target_t_" + k] = targets_t[k]
    for k, v in lt.items():
        out["loss_train_" + k] = torch.as_tensor(v, dtype=torch.float32).reshape(-1)
    G.ONLY.clear()
    G.npz("train_outputs.npz", **out)


if __name__ == "__main__":
    main()
