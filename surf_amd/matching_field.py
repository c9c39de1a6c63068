"""Host-side mirror of models/modules/matching_field.py MatchingField (no parameters)."""
import torch
import torch.nn as nn

from . import ops


class MatchingField(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.n_samples_depths = [int(v) for v in confs.get_list("n_samples_depths")]
        self.n_importance_depths = confs.get_list("n_importance_depths")
        self.up_sample_steps = confs.get_list("up_sample_steps")
        self.depth_res_levels = [int(v) for v in confs.get_list("depth_res_levels")]
        # False (default): the train-mode jitter is drawn like the reference - torch.rand on the CPU generator, then moved
        # (matching_field.py:34) - so that seeded runs reproduce its numbers; True: drawn on the device generator (same
        # distribution, no host work: ~0.1-0.2 s per step at 576x800 on the CPU path)
        self.device_jitter = False

    def draw_jitter(self, nv, n_rays, n_bands, src_idx=0):
        """The train-mode draws of depth_render (:33-35) on the CPU generator, in the reference's order: views in order, only
        the reference view and `src_idx` are perturbed (:129-133), one `torch.rand([batch, 1]) - 0.5` per band."""
        jit = torch.zeros(nv, n_rays, 2)
        for i in range(nv):
            if i == 0 or i == src_idx:
                draws = [torch.rand([n_rays, 1]) - 0.5 for _ in range(n_bands)]        # the reference's draw order
                jit[i, :, :n_bands] = torch.cat(draws, dim=1)                           # one contiguous copy per view
        return jit

    def forward(self, cams, near_fars, hw, matching_volume, stage_idx, range_ratios, pre_depths=None, return_lr=False,
                perturb=False, src_idx=0, saved=None):
        """matching_field.py:73-141 -> depth maps (nv,H,W).  perturb (train mode, surf.py:139): the per-ray z jitter of the
        reference and source views; `occ_reg` (unused by the loss) is not produced."""
        H, W = hw
        jitter = None
        if perturb:
            lvl = self.depth_res_levels[stage_idx]
            n_rays, n_bands = (H // lvl) * (W // lvl), 1 if pre_depths is None else 2
            if self.device_jitter:
                jitter = torch.zeros(cams.nv, n_rays, 2, device=matching_volume.device)
                for i in sorted({0, src_idx}):
                    jitter[i, :, :n_bands] = torch.rand(n_rays, n_bands, device=matching_volume.device) - 0.5
            else:
                jitter = self.draw_jitter(cams.nv, n_rays, n_bands, src_idx).to(matching_volume.device).contiguous()
        if saved is not None:
            saved["jitter"] = jitter
        return ops.matching_depth(matching_volume, cams, near_fars, H, W, self.depth_res_levels[stage_idx],
                                  self.n_samples_depths[stage_idx], pre_depths, range_ratios[stage_idx],
                                  range_ratios[stage_idx - 1] if stage_idx > 0 else 1.0, return_lr=return_lr, jitter=jitter,
                                  saved=saved)          # + "stats": the per-ray softmax statistics for the backward

    def backward(self, cams, near_fars, hw, matching_volume, stage_idx, range_ratios, g_full, pre_depths=None, jitter=None,
                 dmvol=None, src_idx=0, stats=None):
        """d loss / d matching volume from g_full (nv,H,W) = d loss / d this stage's depth maps (zero for the views rendered
        under no_grad, matching_field.py:132); accumulates into `dmvol` if given."""
        H, W = hw
        return ops.matching_depth_backward(matching_volume, cams, near_fars, H, W, self.depth_res_levels[stage_idx],
                                           self.n_samples_depths[stage_idx], g_full, pre_depths, range_ratios[stage_idx],
                                           range_ratios[stage_idx - 1] if stage_idx > 0 else 1.0, jitter=jitter, dmvol=dmvol,
                                           views=(0, src_idx), stats=stats)
