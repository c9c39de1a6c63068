"""Host-side mirror of models/modules/matching_field.py MatchingField (no parameters)."""
import torch.nn as nn

from . import ops


class MatchingField(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.n_samples_depths = [int(v) for v in confs.get_list("n_samples_depths")]
        self.n_importance_depths = confs.get_list("n_importance_depths")
        self.up_sample_steps = confs.get_list("up_sample_steps")
        self.depth_res_levels = [int(v) for v in confs.get_list("depth_res_levels")]

    def forward(self, cams, near_fars, hw, matching_volume, stage_idx, range_ratios, pre_depths=None, return_lr=False):
        """matching_field.py:73-141 with perturb False -> depth maps (nv,H,W)."""
        H, W = hw
        return ops.matching_depth(matching_volume, cams, near_fars, H, W, self.depth_res_levels[stage_idx],
                                  self.n_samples_depths[stage_idx], pre_depths, range_ratios[stage_idx],
                                  range_ratios[stage_idx - 1] if stage_idx > 0 else 1.0, return_lr=return_lr)
