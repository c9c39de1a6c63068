"""Host-side mirror of models/modules/volume.py Volume: holds ``agg_mlp`` (reference key names) and drives the
volume-build kernels (csrc/volume.hip).  Voxel coordinates are int32 triples on a cubic lattice."""
import numpy as np
import torch
import torch.nn as nn

from . import ops
from .grads import accumulate


class Volume(nn.Module):
    def __init__(self, confs):
        super().__init__()
        dims = confs.get_list("base_volume_dim")
        if len(set(dims)) != 1:
            raise NotImplementedError("only cubic base_volume_dim is supported (the reference's sparse lookup "
                                      "projector.py:336-351 is itself only correct for cubic grids)")
        if confs.get("bounding", None) not in (None, [[-1, 1], [-1, 1], [-1, 1]]):
            raise NotImplementedError("only the default bounding box [-1,1]^3 is supported (projector.py:229 hard-codes it)")
        self.base_volume_dim = int(dims[0])
        self.agg_mlp = nn.Sequential(nn.Linear(4, 8), nn.ELU(inplace=True), nn.Linear(8, 1))

    def agg_host(self):
        """The 49 floats of agg_mlp (w1 | b1 | w2 | b2) the cost-volume kernels take by value.  ONE device-to-host copy per
        parameter version (the kernels of all four stages, forward and backward, share it), not four per call."""
        ps = (self.agg_mlp[0].weight, self.agg_mlp[0].bias, self.agg_mlp[2].weight, self.agg_mlp[2].bias)
        key = tuple((p._version, p.data_ptr()) for p in ps)
        if getattr(self, "_agg_cache", None) is None or self._agg_cache[0] != key:
            flat = torch.cat([p.detach().reshape(-1).float() for p in ps])
            self._agg_cache = (key, flat.to("cpu").numpy().copy())
        return self._agg_cache[1]

    def stage_inputs(self, stage, D, feats_c2f, cams, parents=None, parent_feats=None, depths=None, depth_range=None, saved=None):
        """up_sample + depth_filtering + back_proj_multiscale + the row selections of surf.py:97-109.
        Returns coords (N,3) int32 and the U-Net input rows (N, 8 or 16).  saved (dict, train mode): receives `pidx`, the
        child-candidate index of every row (row i copies the features of parent row pidx[i] >> 3), for `stage_backward`."""
        agg = self.agg_host()
        if stage == 0:
            c_all, cv, keep = ops.costvol(feats_c2f, 0, D, cams, agg)
            idx1 = None
        else:
            flags = ops.upsample_filter(parents, D, depths, cams, depth_range)
            idx1 = ops.compact(flags)
            if idx1.shape[0] == 0:
                raise RuntimeError(f"stage {stage}: depth filtering removed every voxel")
            c_all, cv, keep = ops.costvol(feats_c2f, stage, D, cams, agg, parents=parents, idx=idx1)
        idx2 = ops.compact(keep)
        if idx2.shape[0] == 0:
            raise RuntimeError(f"stage {stage}: no voxel is visible in more than one view")
        coords = ops.gather_rows(c_all, idx2)
        reg_in = torch.empty(idx2.shape[0], 8 if stage == 0 else 16, dtype=torch.float32, device=coords.device)
        ops.gather_rows(cv, idx2, dst=reg_in, dst_off=0)
        if stage > 0:
            pidx = ops.compose_index(idx1, idx2)
            ops.gather_rows(parent_feats, pidx, shift=3, dst=reg_in, dst_off=8)
            if saved is not None:
                saved["pidx"] = pidx
        return coords, reg_in

    def stage_backward(self, stage, D, feats_c2f, gfeats_c2f, cams, coords, d_reg_in, g_agg, pidx=None, n_parents=0):
        """Backward of stage_inputs for the kept rows: d_reg_in (N, 8 or 16) -> the FPN maps' gradients (accumulated into
        gfeats_c2f), agg_mlp's (accumulated into g_agg, 49 floats) and, for stage > 0, the gradient of the previous stage's
        `mid` rows (returned, (n_parents, 8)).  The voxel selection (depth filter, visibility) is not differentiable."""
        g_cv = d_reg_in[:, :8].contiguous()
        agg = self.agg_host()
        if ops.side.active("costvol") and d_reg_in.is_cuda:
            # a LEAF of the sweep: its results (the FPN maps' gradients, agg_mlp's 49 floats: float atomics) are read by the FPN
            # backward at the very end; only the parent-feature scatter below feeds the next stage - so the cost-volume backward
            # (4.4 ms a step) leaves the main chain for a side stream (ops.SideStream lane COSTVOL_LANE; SuRF._backward_volumes joins it)
            ops.side.run(lambda: ops.costvol_backward(feats_c2f, gfeats_c2f, stage, D, cams, agg, coords, g_cv, g_agg), lane=ops.COSTVOL_LANE,
                         keep=(g_cv, coords))
        else:
            ops.costvol_backward(feats_c2f, gfeats_c2f, stage, D, cams, agg, coords, g_cv, g_agg)
        if stage == 0:
            return None
        d_mid = torch.zeros(n_parents, 8, dtype=torch.float32, device=d_reg_in.device)
        return ops.scatter_rows_add(d_reg_in, pidx, d_mid, shift=3, dst_off=8)

    def assign_agg_grad(self, g_agg, sink=None):
        """Split the 49 floats of costvol_backward (w1 | b1 | w2 | b2) into agg_mlp's `.grad` (accumulating) or `sink`."""
        parts = ((self.agg_mlp[0].weight, 0, 32), (self.agg_mlp[0].bias, 32, 40), (self.agg_mlp[2].weight, 40, 48),
                 (self.agg_mlp[2].bias, 48, 49))
        for p_, a, b in parts:
            accumulate(p_, g_agg[a:b], sink)
