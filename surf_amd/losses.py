"""Host-side mirror of the reference's training loss (models/losses/loss.py:10-111).

Same constructor keys (`confs/surf.conf:49-63`), same `forward(preds, targets, step, mode)` and the same output dictionary.
Most terms are scalar reductions of the hot path's outputs (plain torch ops: autograd differentiates them); the two with
real work run in HIP and carry their own backward (`surf_amd.autograd`): the local NCC of the surface patches
(`compute_LNCC2`, losses/ncc.py:7-51 -> `surf_lncc` / `surf_lncc_backward`, csrc/lncc.hip) and the per-stage photometric
term (`compute_ptloss`, losses/photometric_loss.py:54-125: inverse warping of the source images by the matching-field depths,
SSIM + smooth-L1 + gradient -> `surf_ptloss_terms` / `surf_ptloss_backward`, csrc/ptloss.hip).  `loss["loss"].backward()` on
the outputs of a train-mode `SuRF.forward` therefore fills every `.grad` (runner.py:158-163); the reference's own `Loss`
(torch throughout) works on those outputs as well.
"""
import torch
import torch.nn as nn

from . import autograd, ops


class Loss(nn.Module):
    KEYS = ("color_weight", "sparse_scale_factor", "sparse_weight", "igr_weight", "mfc_weight", "smooth_weight", "depth_weight",
            "ptloss_weight", "pseudo_auxi_depth_weight", "pseudo_sdf_weight", "pseudo_depth_weight")

    def __init__(self, confs):
        super().__init__()
        for k in self.KEYS:
            setattr(self, k, confs.get_float(k))
        self.stage_weights = confs.get_list("stage_weights")

    @staticmethod
    def _masked_l1(pred, target, mask):
        """sum(|pred - target| mask) / (sum(mask) + 1e-8); mask: a tensor, or "target>0".  On the GPU one fused launch each way
        (autograd.masked_l1) when the three have the same size; the torch expression otherwise."""
        if torch.is_tensor(pred) and pred.is_cuda and pred.shape == target.shape and pred.numel() > 0 and \
                (isinstance(mask, str) or mask.shape == pred.shape):
            return autograd.masked_l1(pred, target.to(pred.device), mask)
        mask = (target > 0).float() if isinstance(mask, str) else mask.float()
        return ((pred - target).abs() * mask).sum() / (mask.sum() + 1e-8)

    def forward(self, preds, targets, step=None, mode="train"):
        # The 2 n photometric terms (5 ms of a step, both directions) depend on the depth maps and the images alone.  Their graph node
        # is created FIRST and on the matching chain's stream (ops.SideStream lane 2): forward, its launches run beside the other
        # terms below; backward, autograd runs the node on that same stream (its stream rule) and orders the gradients that cross
        # streams - so the photometric backward and the matching chain it feeds (through autograd._DepthTap, which lives on the
        # lane too) never occupy the main stream, where the render backward starts at once.
        photos, lane = None, None
        if mode == "train":
            n = len(self.stage_weights)
            imgs_t4 = ops.pack_texel4(targets["imgs"].float().contiguous())
            cams = ops.Cameras(targets["intrs"], targets["c2ws"])
            src_idx = int(targets["src_idx"])
            mask_ref, mask_src = targets["mask_ref"].float().contiguous(), targets["mask_src"].float().contiguous()
            depth_maps = [preds[f"depth_stage{i}"] for i in range(n)] + [preds[f"depth_src_stage{i}"] for i in range(n)]
            specs = [(mask_ref, 0, 2)] * n + [(mask_src, src_idx, 1)] * n
            if imgs_t4.is_cuda and ops.side.active("match") and ops.lane_nodes:
                cur = torch.cuda.current_stream()
                lane = ops.side.lane_stream(2, imgs_t4.device)
                lane.wait_stream(cur)
                with torch.cuda.stream(lane):
                    photos = autograd.photometric_losses(depth_maps, imgs_t4, cams, specs)
            else:
                photos = autograd.photometric_losses(depth_maps, imgs_t4, cams, specs)
        valid_mask = preds["valid_mask"]
        if "mask" in targets:
            valid_mask = valid_mask * targets["mask"].reshape(-1, 1)
        vm = valid_mask.float()
        color_loss = ((preds["color_fine"] - targets["color"]).abs() * vm).sum() / (vm.sum() + 1e-5)       # loss.py:32-33
        eikonal_loss = preds["gradient_error"].mean()
        anneal = min(1.0, step / 2)                                                                          # loss.py:37
        sparse_loss = torch.exp(-preds["sparse_sdf"].abs() * self.sparse_scale_factor).mean() * anneal
        smooth_loss = preds["smooth_error"].mean()
        ncc = preds["ncc"] if "ncc" in preds else autograd.lncc(preds["ref_gray_val"], preds["sampled_gray_val"])
        ncc_mask = vm * preds["mid_inside_sphere"]
        mfc_loss = 0.5 * ((ncc * ncc_mask).sum(dim=0) / (ncc_mask.sum(dim=0) + 1e-8)).squeeze(-1)

        zero = 0.0
        photo_loss = pseudo_auxi = auxi = auxi0 = src_auxi = src_auxi0 = zero
        if mode == "train":
            if lane is not None:                       # the photometric values are read on this stream from here on
                cur = torch.cuda.current_stream()
                cur.wait_stream(lane)
                for ph in photos:
                    ph.record_stream(cur)
            for i in range(n):
                ref_photo, src_photo = photos[i], photos[n + i]
                photo_loss = photo_loss + (ref_photo + src_photo) * self.stage_weights[i]
                pa = self._masked_l1(preds[f"depth_stage{i}"], targets["pseudo_depth_ref"], "target>0")
                spa = self._masked_l1(preds[f"depth_src_stage{i}"], targets["pseudo_depth_src"], "target>0")
                pseudo_auxi = pseudo_auxi + (pa + spa) * self.stage_weights[i]
            auxi = self._masked_l1(preds[f"depth_stage{n - 1}"], targets["depth_ref"], targets["mask_ref"])
            src_auxi = self._masked_l1(preds[f"depth_src_stage{n - 1}"], targets["depth_src"], targets["mask_src"])
            auxi0 = self._masked_l1(preds["depth_stage0"], targets["depth_ref"], targets["mask_ref"])
            src_auxi0 = self._masked_l1(preds["depth_src_stage0"], targets["depth_src"], targets["mask_src"])

        pseudo_sdf_loss = preds["pseudo_sdf"].abs().mean() if "pseudo_sdf" in preds else zero
        pseudo_depth_loss = (self._masked_l1(preds["render_depth"], targets["pseudo_depth"], "target>0")
                             if "pseudo_depth" in targets else zero)
        depth_loss = (self._masked_l1(preds["render_depth"], targets["depth"], "target>0")
                      if "depth" in targets else zero)
        loss = (color_loss * self.color_weight + eikonal_loss * self.igr_weight + sparse_loss * self.sparse_weight
                + mfc_loss * self.mfc_weight + smooth_loss * self.smooth_weight + depth_loss * self.depth_weight
                + photo_loss * self.ptloss_weight + pseudo_auxi * self.pseudo_auxi_depth_weight
                + pseudo_sdf_loss * self.pseudo_sdf_weight + pseudo_depth_loss * self.pseudo_depth_weight)
        return {"loss": loss, "color_loss": color_loss, "eikonal_loss": eikonal_loss, "sparse_loss": sparse_loss,
                "mfc_loss": mfc_loss, "smooth_loss": smooth_loss, "depth_loss": depth_loss, "photo_loss": photo_loss,
                "auxi_depth_loss": auxi, "pseudo_auxi_depth_loss": pseudo_auxi, "src_auxi_depth_loss": src_auxi,
                "pseudo_sdf_loss": pseudo_sdf_loss, "auxi_depth_loss0": auxi0, "src_auxi_depth_loss0": src_auxi0,
                "pseudo_depth_loss": pseudo_depth_loss}
