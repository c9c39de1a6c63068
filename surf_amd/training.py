"""One optimisation step of the reference's runner (runner.py:150-166), driven by the HIP backward kernels.

`finetune_step`: a has_vol model (surf.py:36-45's finetune parameter set): forward (`SuRF.forward("train")`), the loss's
scalar terms through torch autograd on the per-ray outputs, `SuRF.backward` (surf_composite_backward -> surf_sdf_backward /
surf_blend_backward), optimiser step.
`train_step`: a volume-building model (generalisation training): the same, plus the per-stage depth terms (the masked L1
terms through torch autograd on the depth maps, the photometric term through `surf_ptloss_backward`) and
`SuRF.backward_volumes` (matching field -> densify -> sparse U-Net -> cost volume -> FPN), so that every parameter group of
surf.py:36-45 receives its gradient.

Differentiated terms: every term of losses/loss.py - colour, eikonal, sparse-SDF, smooth (H.1), rendered-depth, pseudo-SDF (the dataset's `pseudo_pts`), the patch-NCC
term (`mfc_loss`), the per-stage photometric and pseudo-depth terms.
"""
import torch

from . import dist, ops


def _sync_gradients(optimizer):
    """Data-parallel step (one process per GPU, runner.py's DDP): average the gradients over the ranks with bucketed
    all-reduces (RCCL on GPUs, gloo in the CPU tests) before the optimiser step.  No-op for a single process."""
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        dist.all_reduce_gradients([p for g in optimizer.param_groups for p in g["params"]])

LEAVES = ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "ncc", "smooth_error")


def _leaf_names(preds):
    return LEAVES + (("pseudo_sdf",) if "pseudo_sdf" in preds else ())


def finetune_step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio=1.0, step=0):
    preds = model("train", ipts, cos_anneal_ratio, step)
    preds["ncc"] = ops.lncc(preds["ref_gray_val"].contiguous(), preds["sampled_gray_val"].contiguous())
    leaves = {k: preds[k].detach().clone().requires_grad_(True) for k in _leaf_names(preds)}
    with torch.enable_grad():
        out = loss_fn({**preds, **leaves}, targets, step=step, mode="finetune")      # any mode but "train": no per-stage terms
        out["loss"].backward()
    optimizer.zero_grad(set_to_none=True)
    g = {k: v.grad for k, v in leaves.items()}
    model.backward(g["color_fine"], g["render_depth"], 0.0 if g["gradient_error"] is None else float(g["gradient_error"]),
                   g["sparse_sdf"], g["ncc"], 0.0 if g["smooth_error"] is None else float(g["smooth_error"]), g.get("pseudo_sdf"))
    _sync_gradients(optimizer)
    optimizer.step()
    return {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in out.items()}


def train_step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio=1.0, step=0):
    """runner.py:150-166 for a volume-building model in train mode (model.train(): BatchNorm batch statistics, matching-field
    jitter).  targets: what losses/loss.py reads in mode "train" (color, imgs, intrs, c2ws, src_idx, mask_ref / mask_src,
    pseudo_depth_ref / pseudo_depth_src, depth_ref / depth_src, ...)."""
    if model.has_vol:
        raise ValueError("train_step drives a volume-building model; use finetune_step for has_vol models")
    preds = model("train", ipts, cos_anneal_ratio, step, record=True)
    preds["ncc"] = ops.lncc(preds["ref_gray_val"].contiguous(), preds["sampled_gray_val"].contiguous())
    n = model.num_stage
    depth_keys = [f"depth_stage{i}" for i in range(n)] + [f"depth_src_stage{i}" for i in range(n)]
    leaves = {k: preds[k].detach().clone().requires_grad_(True) for k in _leaf_names(preds) + tuple(depth_keys)}
    with torch.enable_grad():
        out = loss_fn({**preds, **leaves}, targets, step=step, mode="train")
        out["loss"].backward()
    optimizer.zero_grad(set_to_none=True)
    g = {k: v.grad for k, v in leaves.items()}
    # the photometric term is computed by HIP kernels outside autograd: its gradient w.r.t. the depth maps comes from
    # surf_ptloss_backward, weighted like loss.py:60-66
    imgs_t4 = ops.pack_texel4(targets["imgs"].float().contiguous())
    cams = ops.Cameras(targets["intrs"], targets["c2ws"])
    src_idx = int(targets["src_idx"])
    mask_ref, mask_src = targets["mask_ref"].float().contiguous(), targets["mask_src"].float().contiguous()
    g_depths = {}
    for i in range(n):
        w = float(loss_fn.ptloss_weight) * float(loss_fn.stage_weights[i])
        g_ref, g_src = g[f"depth_stage{i}"], g[f"depth_src_stage{i}"]
        if w != 0.0:
            p_ref = ops.photometric_loss_backward(preds[f"depth_stage{i}"].float().contiguous(), imgs_t4, mask_ref, cams, 0, 2, w)
            p_src = ops.photometric_loss_backward(preds[f"depth_src_stage{i}"].float().contiguous(), imgs_t4, mask_src, cams,
                                                  src_idx, 1, w)
            g_ref = p_ref if g_ref is None else g_ref + p_ref
            g_src = p_src if g_src is None else g_src + p_src
        g_depths[i] = (g_ref, g_src)
    rows = model.backward(g["color_fine"], g["render_depth"], 0.0 if g["gradient_error"] is None else float(g["gradient_error"]),
                          g["sparse_sdf"], g["ncc"], 0.0 if g["smooth_error"] is None else float(g["smooth_error"]),
                          g.get("pseudo_sdf"))
    model.backward_volumes(rows, g_depths)
    _sync_gradients(optimizer)
    optimizer.step()
    return {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in out.items()}
