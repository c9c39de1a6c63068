"""One optimisation step of the reference's finetune mode (runner.py:150-166 with a has_vol model), driven by the HIP
backward kernels: forward (`SuRF.forward("train")`), the loss's scalar terms through torch autograd on the per-ray outputs,
`SuRF.backward` (surf_composite_backward -> surf_sdf_backward / surf_blend_backward), optimiser step.

Differentiated terms: colour, eikonal, sparse-SDF, rendered-depth and the patch-NCC term (`mfc_loss`).  NOT differentiated
yet: the smooth term (H.1, weight 1e-4) and, for a model that builds its volumes (has_vol = False), everything upstream of
the feature rows (sparse U-Net, cost volume, FPN) - so only the finetune parameter set of surf.py:36-45 is trained here.
"""
import torch

from . import ops

LEAVES = ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "ncc")


def finetune_step(model, ipts, targets, loss_fn, optimizer, cos_anneal_ratio=1.0, step=0):
    preds = model("train", ipts, cos_anneal_ratio, step)
    preds["ncc"] = ops.lncc(preds["ref_gray_val"].contiguous(), preds["sampled_gray_val"].contiguous())
    leaves = {k: preds[k].detach().clone().requires_grad_(True) for k in LEAVES}
    with torch.enable_grad():
        out = loss_fn({**preds, **leaves}, targets, step=step, mode="finetune")      # any mode but "train": no per-stage terms
        out["loss"].backward()
    optimizer.zero_grad(set_to_none=True)
    g = {k: v.grad for k, v in leaves.items()}
    model.backward(g["color_fine"], g["render_depth"], 0.0 if g["gradient_error"] is None else float(g["gradient_error"]),
                   g["sparse_sdf"], g["ncc"])
    optimizer.step()
    return {k: (float(v.detach()) if torch.is_tensor(v) else float(v)) for k, v in out.items()}
