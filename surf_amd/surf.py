"""Drop-in for models/surf.py SuRF: same constructor, parameter names, ``forward(mode, ipts, cos_anneal_ratio,
step)`` and output keys; every stage runs in the HIP kernels of libsurf_hip.so.

Inference semantics (``mode == "val"`` or a no-grad ``"train"`` forward without the loss-only outputs).  Not
implemented (they belong to SURVEY rows f2/f3): autograd through the kernels, the train-mode jitter of the matching
field (``perturb=True`` in ``build_volumes``), finetune volumes (``has_vol`` / ``init_volumes`` / ``load_params_vol``).
``render.perturb > 0`` (the per-ray jitter of ``ImplicitSurface.render``, active in ``val`` too) is supported.
"""
import torch
import torch.nn as nn

from . import ops
from .feature_network import FeatureNetwork
from .implicit_surface import ImplicitSurface, SceneVolumes
from .matching_field import MatchingField
from .reg_network import SparseCostRegNetList
from .volume import Volume


class SuRF(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.has_vol = confs.get_bool("has_vol", default=False)
        if self.has_vol:
            raise NotImplementedError("has_vol (per-scene finetuning, surf.py:47-78) is not implemented")
        self.range_ratios = [float(r) for r in confs.get_list("range_ratios")]
        self.num_stage = len(self.range_ratios)
        if self.num_stage != 4:
            raise NotImplementedError("the kernels are built for the 4-stage pyramid of confs/*.conf")
        self.feature_network = FeatureNetwork(confs["feature_network"])
        self.volume = Volume(confs["volume"])
        self.reg_network = SparseCostRegNetList(confs["reg_network"])
        self.matching_field = MatchingField(confs["matching_field"])
        self.match_feature_network = FeatureNetwork(confs["feature_network"])
        for p in self.match_feature_network.parameters():
            p.requires_grad = False
        self.implicit_surface = ImplicitSurface(confs["implicit_surface"])

    def get_optim_params(self, lr_conf):
        """surf.py:36-45 (parameter groups; training itself is row f2)."""
        groups = [{"params": list(self.implicit_surface.parameters()), "lr": lr_conf["mlp_lr"]}]
        feat = list(self.feature_network.parameters()) + list(self.reg_network.parameters()) + list(self.volume.parameters())
        groups.append({"params": feat, "lr": lr_conf["feat_lr"]})
        return groups

    @torch.no_grad()
    def build_volumes(self, ipts, features_c2f, cams=None, logit_override=None, timings=None, trace=None):
        """surf.py:80-131 (perturb False).  features_c2f: texel4 maps coarse -> fine.
        Returns (outputs, volumes, tables, matching_volume) with per-stage lists coarse -> fine.
        `logit_override(coords, D) -> (N,)` (bench / tests only) replaces the U-Net's matching logit so that an
        untrained network still produces a realistic, surface-concentrated pyramid; `timings` (dict) receives
        per-stage HIP-event pairs; `trace` (dict, tests only) receives every stage's intermediate tensors."""
        intrs, c2ws = ipts["intrs"], ipts["c2ws"]
        if cams is None:
            cams = ops._cams_ext(ops.Cameras(intrs, c2ws), intrs, c2ws)
        base_range = float((ipts["far"] - ipts["near"]).reshape(-1)[0])
        H, W = ipts["imgs"].shape[-2:]
        D = self.volume.base_volume_dim
        outputs, volumes, tables = {}, [], []
        depths, mvol, coords, mid = None, None, None, None
        src_idx = int(ipts["src_idx"]) if "src_idx" in ipts else 0
        for s in range(self.num_stage):
            if s > 0:
                D *= 2
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if timings is not None else None
            if ev: ev[0].record()
            parents, pre_depths = coords, depths
            coords, reg_in = self.volume.stage_inputs(s, D, features_c2f, cams, coords, mid, depths,
                                                      base_range * self.range_ratios[s])
            if ev: ev[1].record()
            table = ops.table_from_coords(coords, D)
            out, mid = self.reg_network(reg_in, coords, D, s, table=table)
            if logit_override is not None:
                out[:, 0] = logit_override(coords, D)
            if ev: ev[2].record()
            mvol, table = ops.densify(coords, out, D, mvol)
            if ev: ev[3].record()
            depths = self.matching_field(cams, ipts["near_fars"], (H, W), mvol, s, self.range_ratios, depths,
                                         return_lr=trace is not None)
            if trace is not None:
                depths, lr = depths
                trace[s] = {"D": D, "parents": parents, "pre_depths": pre_depths, "coords": coords, "reg_in": reg_in,
                            "out": out, "mvol": mvol, "table": table, "depths": depths, "depths_lr": lr,
                            "depth_range": base_range * self.range_ratios[s]}
            if ev:
                ev[4].record()
                timings[s] = {"n_voxels": int(coords.shape[0]), "events": ev}
            volumes.append(out)          # rows [logit | 7 feature channels]; the SDF kernel reads channels 1..7
            tables.append(table)
            outputs[f"depth_stage{s}"] = depths[0]
            outputs[f"depth_src_stage{s}"] = depths[src_idx]
        return outputs, volumes, tables, mvol

    @torch.no_grad()
    def forward(self, mode, ipts, cos_anneal_ratio=1.0, step=None):
        imgs = ipts["imgs"]
        intrs, c2ws = ipts["intrs"], ipts["c2ws"]
        cams = ops._cams_ext(ops.Cameras(intrs, c2ws), intrs, c2ws)
        features = self.feature_network(imgs)                                   # texel4, coarse -> fine
        outputs, volumes, tables, mvol = self.build_volumes(ipts, features, cams)
        # the second (frozen) FPN pass of surf.py:147-148 only feeds the loss-only patch warp: skipped
        scene = SceneVolumes.from_device_layouts(mvol, [v[:, 1:] for v in volumes[::-1]], tables[::-1], features[::-1],
                                                 ops.pack_texel4(imgs.detach().float().contiguous()), cams)
        isurf = self.implicit_surface
        rays_o, rays_d = ipts["rays_o"], ipts["rays_d"]
        near, far = ipts["near"], ipts["far"]
        if near.shape[0] == 1:
            near = near.repeat(rays_o.shape[0], 1)
            far = far.repeat(rays_o.shape[0], 1)
        if mode == "val":
            surface = isurf.validate(rays_o, rays_d, near, far, scene, ipts["bound_min"], ipts["bound_max"], ipts["hw"],
                                     cos_anneal_ratio, step, mesh_resolution=int(ipts.get("mesh_resolution", 512)))
        else:
            surface = isurf.render_scene(rays_o, rays_d, near, far, scene, cos_anneal_ratio)
        outputs.update(surface)
        return outputs
