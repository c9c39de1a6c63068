"""Host-side mirror of the reference's implicit-surface stage (models/modules/implicit_surface.py).

Same module / parameter names (so reference checkpoints load with ``load_state_dict``), same
``render`` / ``validate`` / ``extract_geometry`` / ``forward`` entry points and output keys; the
arithmetic runs in the HIP kernels behind ``surf_amd.ops`` (no PyTorch fallback).

Scope of this mirror: inference (`val`) semantics, including the ``render.perturb`` jitter, and the
forward side of a training step: ``render_scene(patch_warp=True)`` adds the loss-only outputs of ``render_core``
(``ref_gray_val`` / ``sampled_gray_val`` patch warps, ``smooth_error``, ``sparse_sdf``).  There is no autograd graph: the
backward of a training forward is ``backward_render``, which chains the HIP backward kernels for given
gradients of the outputs.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .grads import accumulate


class _WNLinear(nn.Module):
    """Parameter container with ``nn.utils.weight_norm``'s names: weight_g (out,1), weight_v (out,in), bias."""

    def __init__(self, weight, bias):
        super().__init__()
        self.weight_g = nn.Parameter(torch.linalg.norm(weight, dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(weight.clone())
        self.bias = nn.Parameter(bias.clone())


class SDFNetworkSparse(nn.Module):
    """sdf_network.py:27-93: parameters + geometric initialisation.  Evaluation: ops.sdf_mlp."""

    def __init__(self, d_in=3, d_out=129, d_hidden=128, n_layers=6, skip_in=(3,), multires=4, bias=0.5, scale=1.0,
                 geometric_init=True, weight_norm=True, inside_outside=False, feat_channels=28, feat_multires=0):
        super().__init__()
        if not (d_in == 3 and d_hidden == 128 and n_layers == 6 and tuple(skip_in) == (3,) and multires == 4
                and feat_channels == 28 and feat_multires == 0 and float(scale) == 1.0 and weight_norm):
            raise NotImplementedError("the HIP SDF kernel implements the architecture shipped in confs/*.conf "
                                      "(d_hidden 128, n_layers 6, skip_in [3], multires 4, feat_channels 28, scale 1)")
        self.scale = scale
        in0 = 3 + 3 * 2 * multires
        dims = [in0] + [d_hidden + feat_channels] * n_layers + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in skip_in else dims[l + 1]
            if l < self.num_layers - 2:
                out_dim -= feat_channels
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:  # sdf_network.py:62-86
                with torch.no_grad():
                    if l == self.num_layers - 2:
                        sgn = -1.0 if inside_outside else 1.0
                        nn.init.normal_(lin.weight, mean=sgn * math.sqrt(math.pi) / math.sqrt(dims[l]), std=0.0001)
                        nn.init.constant_(lin.bias, -sgn * bias)
                        lin.weight[:, -feat_channels:] = 0.0
                        lin.bias[-feat_channels:] = 0.0
                    elif l == 0:
                        nn.init.constant_(lin.bias, 0.0)
                        lin.weight[:, 3:] = 0.0
                        nn.init.normal_(lin.weight[:, :3], 0.0, math.sqrt(2) / math.sqrt(out_dim))
                    elif l in skip_in:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.normal_(lin.weight, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                        lin.weight[:, -(dims[0] - 3 + feat_channels):] = 0.0
                    else:
                        nn.init.constant_(lin.bias, 0.0)
                        nn.init.normal_(lin.weight, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                        lin.weight[:, -feat_channels:] = 0.0
            setattr(self, f"lin{l}", _WNLinear(lin.weight.detach(), lin.bias.detach()))


class BlendingNetwork(nn.Module):
    """blending_network.py:22-67: parameters + initialisation.  Evaluation: ops.blend."""

    def __init__(self, d_feature=16, anti_alias_pooling=True):
        super().__init__()
        if d_feature != 16 or not anti_alias_pooling:
            raise NotImplementedError("the HIP blending kernel implements BlendingNetwork(d_feature=16, anti_alias_pooling=True)")
        self.s = nn.Parameter(torch.tensor(0.2))
        act = nn.ELU(inplace=True)
        f = d_feature + 3
        self.ray_dir_fc = nn.Sequential(nn.Linear(4, 16), act, nn.Linear(16, f), act)
        self.base_fc = nn.Sequential(nn.Linear(f * 3, 64), act, nn.Linear(64, 32), act)
        self.vis_fc = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 33), act)
        self.vis_fc2 = nn.Sequential(nn.Linear(32, 32), act, nn.Linear(32, 1), nn.Sigmoid())
        self.rgb_fc = nn.Sequential(nn.Linear(32 + 1 + 4, 16), act, nn.Linear(16, 8), act, nn.Linear(8, 1))
        for seq in (self.base_fc, self.vis_fc2, self.vis_fc, self.rgb_fc):  # blending_network.py:8-12,63-66
            for m in seq:
                if isinstance(m, nn.Linear):
                    nn.init.kaiming_normal_(m.weight.data)
                    nn.init.zeros_(m.bias.data)


class SingleVarianceNetwork(nn.Module):
    """variance_network.py:5-11."""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(float(init_val))))

    def inv_s(self):
        """exp(10 v) clipped, as a python float (the kernels take it by value).  One device read per parameter VERSION, not
        per render call: an inference loop never syncs on it after the first call."""
        key = (self.variance._version, self.variance.data_ptr())
        if getattr(self, "_inv_s_cache", None) is None or self._inv_s_cache[0] != key:
            self._inv_s_cache = (key, float(torch.exp(self.variance.detach() * 10.0).clamp(1e-6, 1e6)))
        return self._inv_s_cache[1]


class SceneVolumes:
    """Device-resident inputs of the renderer for one scene, in the kernels' layouts."""

    def __init__(self, matching_volume, volumes, sparse_idxes, features, imgs, intrs, c2ws):
        dev = imgs.device
        mv = matching_volume
        if mv.dim() == 5:
            mv = mv[0, 0]
        self.mvol = mv.contiguous().float()
        self.sv = ops.SparseVolumes([v.detach().float() for v in volumes], list(sparse_idxes))
        self.feats_t4 = [ops.pack_texel4(f.detach().float().contiguous()) for f in features]  # NCHW -> texel4
        self.imgs_t4 = ops.pack_texel4(imgs.detach().float().contiguous())
        self.cams = ops.Cameras(intrs, c2ws)
        self.device = dev
        self.match_feats_t4 = None
        self._warp_maps = {}

    def occupied_any(self, pts):
        """lookup_volume(pts, mask_volumes, 'nearest').any(-1) (implicit_surface.py:175): is the nearest voxel of any
        level occupied; the mask volume is 1 exactly where the index table is >= 0 (volume.py:112-130)."""
        return ops.occupied_any(pts.float().contiguous(), self.sv)       # one launch (until round 5: 15 torch ops per level)

    def warp_maps(self, use_match=False):
        """implicit_surface.py:229-241: FPN levels 0, 1, 2 at the finest level's size (texel4), from `features` or - once
        training is past step 2 - from the frozen `match_features`; built on first use."""
        key = bool(use_match and self.match_feats_t4 is not None)
        if key and getattr(self, "match_ready", None) is not None:      # launched on a side stream before the volume build
            ops.side.wait_for(self.match_ready)
            self.match_ready = None
        if key not in self._warp_maps:
            f = self.match_feats_t4 if key else self.feats_t4
            H, W = f[0].shape[1:3]
            self._warp_maps[key] = [f[0], ops.upsample_bilinear_t4(f[1], H, W), ops.upsample_bilinear_t4(f[2], H, W)]
        return self._warp_maps[key]

    @classmethod
    def from_device_layouts(cls, mvol, volumes, tables, feats_t4, imgs_t4, cams):
        """Already in kernel layouts (the SuRF pipeline): no re-packing.  volumes: (N,7) or (N,8) rows, fine -> coarse."""
        self = cls.__new__(cls)
        self.mvol = mvol
        self.sv = ops.SparseVolumes(list(volumes), list(tables))
        self.feats_t4 = list(feats_t4)
        self.imgs_t4 = imgs_t4
        self.cams = cams
        self.device = mvol.device
        self.match_feats_t4 = None
        self._warp_maps = {}
        return self


class _LatticeScene:
    """The part of SceneVolumes the SDF lattice needs (no images / cameras): extract_geometry's reference signature."""

    def __init__(self, volumes, sparse_idxes):
        self.sv = ops.SparseVolumes([v.detach().float() for v in volumes], list(sparse_idxes))
        self.device = self.sv.vols[0].device


def _occupied_list(rec, key):
    """The indices of the occupied points of a forward record's boolean mask `key`, computed once (a synchronising read: a
    training forward does it right after the mask, where the host is waiting for the render anyway; see ops.SideStream)."""
    lst = rec.get(key + "_idx")
    if lst is None:
        lst = rec[key + "_idx"] = rec[key].nonzero().squeeze(1)
    return lst


class ImplicitSurface(nn.Module):
    """implicit_surface.py:50-436 (inference semantics)."""

    def __init__(self, confs):
        super().__init__()
        self.n_samples = [int(n) for n in confs.get_list("render.n_samples")]
        self.sample_ranges = [float(r) for r in confs.get_list("render.sample_ranges")]
        self.n_depth = confs.get_int("render.n_depth")
        self.perturb = confs.get_float("render.perturb")
        self.val_chunk = confs.get_int("render.val_chunk", 1 << 19)     # rays per launch of validate (ours; see val_chunk_rays)
        self.sdf_network = SDFNetworkSparse(**dict(confs["sdf_network"]))
        self.color_network = BlendingNetwork(**dict(confs["color_network"]))
        self.deviation_network = SingleVarianceNetwork(**dict(confs["variance_network"]))
        # which SDF kernel evaluates sdf_network (ops.SDF_PRECISIONS): "f32" = fp32 MFMA; "bf16x3" = exact three-way bf16
        # operand split on the bf16 pipe (fp32-equivalent, default); "f16x2" = two fp16 pieces (22-bit operands, fastest)
        self.sdf_precision = confs.get_string("render.sdf_precision", "bf16x3")
        if self.sdf_precision not in ops.SDF_PRECISIONS:
            raise ValueError(f"render.sdf_precision must be one of {ops.SDF_PRECISIONS}, got {self.sdf_precision!r}")
        # which blending kernel evaluates color_network (ops.BLEND_PRECISIONS)
        self.blend_precision = confs.get_string("render.blend_precision", ops.BLEND_DEFAULT)
        if self.blend_precision not in ops.BLEND_PRECISIONS:
            raise ValueError(f"render.blend_precision must be one of {ops.BLEND_PRECISIONS}, got {self.blend_precision!r}")
        self._packed = None
        self.kernel_events = None          # bench.py: list receiving (name, start, end) HIP event triples
        self.active_samples_log = None     # bench.py: list receiving the active-sample count of every render call
        self.last_active_samples = None
        # loading a checkpoint or switching train / eval drops the cached weight re-layouts
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate_packed())

    # ---- weight re-layouts are cached and refreshed whenever parameters change -------------------------
    def invalidate_packed(self):
        """Drop the cached MFMA weight re-layouts.  The cache key is (Parameter._version, data_ptr) of every parameter,
        which optimiser steps, `copy_` on the parameter and `load_state_dict` all change; an in-place write through
        `param.data` (e.g. `p.data.mul_(2)`) changes neither, so call this after such an edit.  Also drops the flat
        effective-weight vector both SDF images are cut from (packing._sdf_flat, cached on the SDF network under the same key)."""
        self._packed = None
        if getattr(self.sdf_network, "_surf_flat", None) is not None:
            object.__setattr__(self.sdf_network, "_surf_flat", None)

    def train(self, mode=True):
        self.invalidate_packed()
        return super().train(mode)

    def packed_weights(self, device):
        """(SDF operand stream, blend LDS image) for the kernels of sdf_precision / blend_precision, cached per parameter version.
        With the split kernels (the defaults) the images are built ON THE DEVICE from the live parameters (surf_amd.packing:
        byte-identical to the C ABI's host packers, no device-to-host copy, no sync); the fp32-MFMA kernels use the host packers."""
        ver = tuple((p._version, p.data_ptr()) for p in self.parameters()) + (str(device), self.sdf_precision, self.blend_precision)
        if self._packed is None or self._packed[0] != ver:
            from . import packing
            on_device = next(self.parameters()).is_cuda and packing.supported(self.sdf_precision, self.blend_precision)
            if on_device:
                with torch.no_grad():
                    sdf_w = packing.sdf_pack_split_device(self.sdf_network, self.sdf_precision)
                    blend_w = packing.blend_pack_split_device(self.color_network, self.blend_precision)
            else:
                sd = {k: v for k, v in self.state_dict().items()}
                sdf_w = (ops.sdf_pack_weights(sd, device, "sdf_network.") if self.sdf_precision == "f32" else
                         ops.sdf_pack_weights_split(sd, device, "sdf_network.", self.sdf_precision))
                blend_w = ops.blend_pack_weights(sd, device, "color_network.", self.blend_precision)
            self._packed = (ver, sdf_w, blend_w)
        return self._packed[1], self._packed[2]

    def smooth_weights(self, device):
        """fp32 image of sdf_network for the second-order kernel (training only), cached with the other re-layouts."""
        self.packed_weights(device)                  # refreshes self._packed when parameters changed
        if len(self._packed) == 3:
            if next(self.parameters()).is_cuda:
                from . import packing
                with torch.no_grad():
                    sm = packing.sdf_pack_smooth_device(self.sdf_network)
            else:
                sm = ops.sdf_smooth_pack_weights({k: v for k, v in self.state_dict().items()}, device, "sdf_network.")
            self._packed = self._packed + (sm,)
        return self._packed[3]

    def scene(self, matching_volume, volumes, sparse_idxes, mask_volumes, features, imgs, intrs, c2ws):
        """mask_volumes are redundant with the index tables (mask == table >= 0, volume.py:112-130) and unused."""
        return SceneVolumes(matching_volume, volumes, sparse_idxes, features, imgs, intrs, c2ws)

    def render_scene(self, rays_o, rays_d, near, far, scene, cos_anneal_ratio=1.0, per_sample=True, jitter=None,
                     patch_warp=False, step=None):
        """render (:268-335) + render_core (:64-266) on prepared SceneVolumes.
        render.perturb > 0 (every shipped conf; the reference jitters even in `val`, :274-277, 304-306): one
        `torch.rand([R, 1]) - 0.5` per stage drawn on the CPU generator in the reference's order, so that a seeded run
        reproduces the reference's sample positions for the same ray batch; `jitter` (R, n_stage) overrides the draw.
        patch_warp: also the training outputs ref_gray_val / sampled_gray_val (:217-245, row a15): the 11x11 homography
        patches of the stacked feature maps around every ray's SDF zero crossing (from match_features once step >= 2)."""
        dev = rays_o.device
        self._random_pts = None
        if jitter is None and self.perturb > 0:
            jitter = self.draw_jitter(rays_o.shape[0])
        if jitter is not None:
            jitter = jitter.to(dev, torch.float32).contiguous()
        sdf_w, blend_w = self.packed_weights(dev)
        rays_o = rays_o.float().contiguous()
        rays_d = rays_d.float().contiguous()
        ev = self.kernel_events                      # optional per-kernel HIP event pairs (bench.py)

        def timed(name, fn):
            if ev is None:
                return fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            r = fn()
            b.record()
            ev.append((name, a, b))
            return r

        st = timed("ray_setup", lambda: ops.ray_setup(rays_o, rays_d, near.float(), far.float(), scene.mvol, scene.sv,
                                                      self.n_samples, self.sample_ranges, self.n_depth, jitter=jitter,
                                                      want_z=patch_warp))
        # masked-in samples, ray-major order.  Inference with the split kernels keeps the count on the device (no host sync
        # between the compaction and the two MLP kernels); a training forward needs the exact-size list for its backward
        n_act = None
        if patch_warp or self.sdf_precision == "f32" or self.blend_precision == "f32":
            act = timed("compact", lambda: ops.compact(st["vmask"]))
            if patch_warp and act.shape[0] == 0:
                # implicit_surface.py:88-89: a training batch without a single masked-in sample still sends its first ten points
                # through the networks (their compositing weights stay zero: voxel_mask is unchanged) - sparse_sdf then holds
                # ten real SDF values and the graph stays connected.  (Inference outputs do not depend on it.)
                act = torch.arange(min(10, st["vmask"].numel()), dtype=torch.int32, device=dev)
        else:
            act, n_act = timed("compact", lambda: ops.compact_counted(st["vmask"]))
        side_fwd = None
        if patch_warp:
            # Two branches of a training forward depend on the sample positions alone and meet the rest only in the output
            # dictionary: smooth_error (:100-103, :172: |H.1| of the SDF averaged over the masked-in samples inside the unit
            # sphere; 1.3 ms on the step's 65 k samples) and the random points of sparse_sdf (:174-178).  On side streams beside
            # the SDF / blend / compositing / patch-warp chain (ops.SideStream), joined before the dictionary is filled.
            def smooth_branch():
                smooth, _ = ops.sdf_smooth(st["pts"], scene.sv, self.smooth_weights(dev), active_idx=act)
                inside = ((torch.linalg.norm(st["pts"], ord=2, dim=-1) < 1.0) & st["vmask"].bool()).float()
                err = (torch.linalg.norm(smooth, ord=2, dim=-1) * inside).sum() / (inside.sum() + 1e-5)
                return smooth, inside, err

            def sparse_branch():
                pr = self._random_pts if self._random_pts is not None else torch.rand([1024, 3]) * 2 - 1
                pr = pr.to(dev, torch.float32).contiguous()
                occ = scene.occupied_any(pr)
                sdf_r, _ = ops.sdf_mlp(pr, scene.sv, sdf_w, mask=occ.to(torch.uint8), want_grad=False)
                return pr, occ, torch.where(occ, sdf_r, torch.zeros_like(sdf_r))

            if ops.side.active("fwd") and dev.type == "cuda" and ev is None:
                self.smooth_weights(dev)                     # packed on this stream, once per parameter version
                side_fwd = (ops.side.run(smooth_branch, lane=1), ops.side.run(sparse_branch, lane=0))
        sdf, grad = timed("sdf_mlp", lambda: ops.sdf_mlp(st["pts"], scene.sv, sdf_w, mask=st["vmask"], active_idx=act,
                                                         active_count=n_act))
        col, nvalid = timed("blend", lambda: ops.blend(st["pts"], scene.feats_t4, scene.imgs_t4, scene.cams, blend_w,
                                                       mask=st["vmask"], active_idx=act, active_count=n_act))
        out = timed("composite", lambda: ops.composite(sdf, grad, col, nvalid, st, rays_d, self.deviation_network.inv_s(),
                                                       float(cos_anneal_ratio), scene.cams, per_sample=per_sample,
                                                       want_z0=patch_warp))
        if ev is not None:                                  # bench.py only: the count for the roofline arithmetic
            self.last_active_samples = int(act.shape[0]) if n_act is None else n_act
            if self.active_samples_log is not None:
                self.active_samples_log.append(self.last_active_samples)
        R, S = st["mid_z"].shape
        eik = out.pop("eik").sum(dim=0)
        out["gradient_error"] = eik[0] / (eik[1] + 1e-5)
        out["valid_mask"] = out["valid_mask"].bool()
        out["mid_inside_sphere"] = out["mid_inside_sphere"].float()
        out["mid_z_vals"] = st["mid_z"]
        if per_sample:
            out["gradients"] = grad.view(R, S, 3)
            out["sdf"] = sdf.view(R, S)
        out["s_val"] = torch.full((1, 1), 1.0 / self.deviation_network.inv_s(), device=dev)
        if patch_warp:
            pts0 = ops.surface_points(rays_o, rays_d, out.pop("z_sdf0"), st["z_vals"])
            _, g0 = ops.sdf_mlp(pts0, scene.sv, sdf_w)                           # :221: gradient at the crossing, unmasked
            maps = scene.warp_maps(use_match=not (step is None or step < 2))
            out["ref_gray_val"], out["sampled_gray_val"] = ops.patch_warp(pts0, g0, maps, scene.cams)
            out["pts_sdf0"], out["gradients_sdf0"] = pts0, g0
            ops.side.join(lanes=(0, 1, 3))                   # (3: the frozen matching FPN's stream, if it is still open)
            if side_fwd is not None:
                (smooth, inside, out["smooth_error"]), (pr, occ, sdf_r0) = side_fwd
            else:
                smooth, inside, out["smooth_error"] = smooth_branch()
                pr, occ, sdf_r0 = sparse_branch()
            # sparse_sdf (:174-178, :255): the SDF at 1024 uniform points (zero where no level is occupied) + at the samples
            out["sparse_sdf"] = torch.cat([sdf_r0, sdf]).view(-1, 1)
            # what backward_render needs of this forward (row f2: the partial backward of the render)
            self._ctx = dict(st=st, act=act, sdf=sdf, grad=grad, col=col, rays_d=rays_d, anneal=float(cos_anneal_ratio),
                             scene=scene, eik_den=float(eik[1]), random_pts=pr, random_occ=occ, pts0=pts0, g0=g0, maps=maps,
                             smooth=smooth, inside=inside)
            _occupied_list(self._ctx, "random_occ")
        return out

    @torch.no_grad()
    def backward_render(self, g_color, g_depth=None, g_gradient_error=0.0, g_sparse_sdf=None, g_ncc=None, gfeats_t4=None,
                        g_smooth_error=0.0, g_pseudo_sdf=None, g_patches=None, ctx=None, sink=None, rows8=False):
        """Partial backward of the last training forward (`render_scene(patch_warp=True)`), SURVEY 8f-f2: given the loss's
        gradients w.r.t. `color_fine` (R,3), `render_depth` (R), `gradient_error` (scalar), `sparse_sdf` ((1024 + R*S),1) and
        the per-ray patch NCC (R,1) (= compute_LNCC2 of ref_gray_val / sampled_gray_val, the mfc term),
        ACCUMULATES `.grad` on every parameter of the implicit surface - sdf_network.lin*.{weight_g, weight_v, bias},
        color_network.*, deviation_network.variance - and returns the gradients of the scene's sparse feature rows, fine ->
        coarse, (N_s, 7).  Kernels: surf_composite_backward -> surf_sdf_backward (reverse over forward: the spatial-gradient
        upstream is a tangent direction), surf_blend_backward, and for the NCC term surf_patch_warp_tangent -> surf_lncc_jvp ->
        surf_crossing_backward (d ncc / d z0 as a forward-mode tangent along the ray, then into the two bracketing samples).
        gfeats_t4 (fine -> coarse, like the scene's feature maps): accumulates the colour path's gradient into the FPN maps
        (generalisation training).  g_gradient_error / g_smooth_error: python floats or 0-d device tensors (autograd's: they are
        then never read back).  g_smooth_error: the smooth (H.1) term, through surf_sdf_smooth_backward (reverse
        over a forward with value, two tangents and their mixed tangent).  g_pseudo_sdf (n_pseudo, 1): the pseudo_sdf output
        (`ImplicitSurface.pseudo_sdf` after the training forward).  The volume build's backward is SuRF.backward_volumes.
        g_patches = (d loss / d ref_gray_val, d loss / d sampled_gray_val) - what autograd hands back when the loss consumed
        the patch stacks themselves (the reference's Loss: compute_LNCC2 in torch, losses/loss.py:43): contracted with the
        patches' tangents along the ray into d loss / d z0 (the patches depend on the network through z0 alone,
        implicit_surface.py:217-245).  ctx: the record of the forward to differentiate (default: the module's last one);
        sink: a grads.GradSink that receives the parameter gradients instead of `.grad` (surf_amd.autograd).
        rows8: return the kernels' own (N_s, 8) rows = [7 features | 0] instead of their (N_s, 7) slices (no copy)."""
        c = ctx if ctx is not None else self._ctx
        st, act, scene = c["st"], c["act"], c["scene"]
        dev = c["sdf"].device
        inv_s = self.deviation_network.inv_s()
        if g_color is None:
            g_color = torch.zeros(c["rays_d"].shape[0], 3, dtype=torch.float32, device=dev)
        # scalar upstream gradients may arrive as python floats or - from autograd - as device scalars, which stay on the device
        eik_up = g_gradient_error if torch.is_tensor(g_gradient_error) else None
        d_sdf, d_grad, d_col, d_is = ops.composite_backward(c["sdf"], c["grad"], c["col"], st, c["rays_d"], inv_s, c["anneal"], scene.cams,
                                                        g_color.float().contiguous(),
                                                        None if g_depth is None else g_depth.float().contiguous(),
                                                        eik_scale=(1.0 if eik_up is not None else float(g_gradient_error)) / (c["eik_den"] + 1e-5),
                                                        eik_upstream=eik_up)
        if g_ncc is not None or g_patches is not None:
            ref, src, ref_t, src_t = ops.patch_warp_tangent(c["pts0"], c["rays_d"], c["g0"], c["maps"], scene.cams)
            g_z0 = torch.zeros(ref.shape[1], dtype=torch.float32, device=dev)
            if g_ncc is not None:
                _, dncc = ops.lncc_jvp(ref, src, ref_t, src_t)
                g_z0 += g_ncc.reshape(-1).float() * dncc
            if g_patches is not None:
                if g_patches[0] is not None:
                    g_z0 += (g_patches[0].float() * ref_t).sum(dim=(0, 2, 3))
                if g_patches[1] is not None:
                    g_z0 += (g_patches[1].float() * src_t).sum(dim=(0, 2, 3))
            ops.crossing_backward(c["sdf"], st["vmask"], st["mid_z"], st["z_vals"].max(), g_z0, d_sdf)

        def blend_branch():
            cn = dict(self.color_network.named_parameters())     # raw parameter buffer in state_dict order, built on the device
            raw_w = torch.cat([cn[k].detach().reshape(-1).float() for k in ops.BLEND_KEYS]).contiguous()
            return ops.blend_backward(st["pts"], act, d_col, scene.feats_t4, scene.imgs_t4, scene.cams, raw_w, gfeats_t4=gfeats_t4)

        # Three branches hang off the compositing backward and meet again only in the parameter gradients / the rows'
        # atomics: colour (surf_blend_backward + its weight-gradient GEMMs: one-wavefront workgroups), smooth (H.1: its layer
        # launches + GEMMs) and the SDF value / gradient backward.  Each works on the step's 65 k samples - latency bound, a
        # fraction of the chip's wave slots - so the first two run on side streams beside the third (ops.SideStream).
        on_side = ops.side.active("render") and dev.type == "cuda"
        idx = act.long()
        ybar = d_sdf[idx]
        pts = st["pts"][idx]
        gbar = d_grad[idx]
        if g_sparse_sdf is not None:
            gs = g_sparse_sdf.reshape(-1).float()
            ybar = ybar + gs[1024:][idx]                      # the samples' own share of sparse_sdf (masked-out rows are constants)
            sel = _occupied_list(c, "random_occ")             # index lists: no synchronising mask reads in the sweep
            pts = torch.cat([pts, c["random_pts"][sel]])
            ybar = torch.cat([ybar, gs[:1024][sel]])
            gbar = torch.cat([gbar, torch.zeros(sel.shape[0], 3, device=dev)])
        if g_pseudo_sdf is not None and "pseudo_pts" in c:        # pseudo_sdf (:425-434): the occupied pseudo points' SDF values
            sel_p = _occupied_list(c, "pseudo_occ")
            pts = torch.cat([pts, c["pseudo_pts"][sel_p]])
            ybar = torch.cat([ybar, g_pseudo_sdf.reshape(-1).float()[sel_p]])
            gbar = torch.cat([gbar, torch.zeros(sel_p.shape[0], 3, device=dev)])
        if on_side:
            gb = ops.side.run(blend_branch, lane=0, keep=(d_col,))
        sw = self.smooth_weights(dev)
        dvols = [torch.zeros_like(v) for v in scene.sv.vols]      # both SDF branches add into these rows with atomics
        rs = None
        if torch.is_tensor(g_smooth_error) or g_smooth_error:
            def smooth_branch():
                # smooth_error = sum_n inside_n |smooth_n| / (sum inside + 1e-5)  ->  sbar_n = g inside_n smooth_n / (|smooth_n| den)
                sm, ins = c["smooth"][idx], c["inside"][idx]
                nrm = torch.linalg.norm(sm, ord=2, dim=-1, keepdim=True)
                g_sm = g_smooth_error.detach().float() if torch.is_tensor(g_smooth_error) else float(g_smooth_error)
                scale = g_sm / (c["inside"].sum() + 1e-5)                         # a device scalar: no host round trip mid-sweep
                sbar = scale * ins[:, None] * sm / nrm.clamp_min(1e-30)
                return ops.sdf_smooth_backward(st["pts"][idx].contiguous(), sbar.contiguous(), scene.sv, sw, dvols=dvols)

            rs = ops.side.run(smooth_branch, lane=ops.SMOOTH_LANE, keep=(idx,)) if on_side else smooth_branch()
        res = ops.sdf_backward(pts.contiguous(), ybar.contiguous(), gbar.contiguous(), scene.sv, sw, dvols=dvols)
        if on_side:                                            # the three branches meet here
            ops.side.join(lanes=(0, ops.SMOOTH_LANE))
        else:
            gb = blend_branch()
        if rs is not None:
            for l in range(7):
                res["weight"][l] = res["weight"][l] + rs["weight"][l]
                res["bias"][l] = res["bias"][l] + rs["bias"][l]
        # weight norm W = g v / |v|_row (sdf_network.py:88-89), its backward in closed form for the seven layers in one launch
        # (ops.weight_norm_backward):  dg = <dW, v>/|v|,  dv = (g/|v|) (dW - (dg/|v|) v)
        lins = [getattr(self.sdf_network, f"lin{l}") for l in range(7)]
        dvs, dgs = ops.weight_norm_backward([lin.weight_v.detach().float().contiguous() for lin in lins],
                                            [lin.weight_g.detach().float().contiguous() for lin in lins],
                                            [res["weight"][l].contiguous() for l in range(7)])
        for l, lin in enumerate(lins):
            accumulate(lin.weight_g, dgs[l], sink)
            accumulate(lin.weight_v, dvs[l], sink)
            accumulate(lin.bias, res["bias"][l], sink)
        for name, p in self.color_network.named_parameters():
            accumulate(p, gb[name], sink)
        var = self.deviation_network.variance
        # inv_s = exp(10 v) clamped to [1e-6, 1e6] (one device read per parameter version): strictly inside = not clamped
        dvar = d_is * 10.0 * inv_s if 1e-6 < inv_s < 1e6 else torch.zeros((), device=dev)
        accumulate(var, dvar, sink)
        return list(res["volumes"]) if rows8 else [g[:, :7].contiguous() for g in res["volumes"]]

    def draw_jitter(self, n_rays, ref_chunk=None):
        """The `torch.rand([batch, 1]) - 0.5` draws of ImplicitSurface.render (:274-277, :304-306) on the CPU generator,
        in the reference's order: per render() call one draw per stage, followed by the `torch.rand([1024, 3])` of
        render_core (:174), which is drawn and dropped here to keep the generator aligned.  `ref_chunk` = the
        reference's batch per render() call: None = one call over all rays (train), 256 = validate (:367-370)."""
        step = n_rays if not ref_chunk else int(ref_chunk)
        cols = []
        for s0 in range(0, n_rays, step):
            b = min(step, n_rays - s0)
            cols.append(torch.cat([torch.rand([b, 1]) - 0.5 for _ in self.n_samples], dim=1))
            self._random_pts = torch.rand([1024, 3]) * 2 - 1          # sparse_sdf's points of a training forward
        return torch.cat(cols, dim=0) if cols else torch.zeros(0, len(self.n_samples))

    def render(self, rays_o, rays_d, near, far, matching_volume, volumes, sparse_idxes, mask_volumes, imgs, features,
               match_features, intrs, c2ws, cos_anneal_ratio, step):
        scene = self.scene(matching_volume, volumes, sparse_idxes, mask_volumes, features, imgs, intrs, c2ws)
        return self.render_scene(rays_o, rays_d, near, far, scene, cos_anneal_ratio)

    def sdf_grid(self, scene, bound_min, bound_max, resolution):
        """The lattice of extract_geometry (:337-351): u[x,y,z] = -sdf, evaluated by the forward-only kernel.  With the split
        kernels (the defaults) the kernel forms every point from its lattice index and the three axis arrays (torch.linspace's
        values, :338-340) and writes -sdf straight into `u`, marching cubes' input: no point tensor (1.6 GB written and re-read at
        512^3 before round 5), no negated copy.  sdf_precision = f32 (the fp32-MFMA kernel) still takes point tensors."""
        dev = scene.device
        sdf_w, _ = self.packed_weights(dev)
        bmin = bound_min.detach().to("cpu", torch.float32)
        bmax = bound_max.detach().to("cpu", torch.float32)
        axes = [torch.linspace(float(bmin[a]), float(bmax[a]), resolution).to(dev) for a in range(3)]
        u = torch.empty(resolution, resolution, resolution, dtype=torch.float32, device=dev)
        lattice_mode = self.sdf_precision != "f32"
        per_launch = (1 << 31) - 1 if lattice_mode else (1 << 24)           # points per launch
        slab = max(1, per_launch // (resolution * resolution))
        for x0 in range(0, resolution, slab):
            nx = min(slab, resolution - x0)
            if self.kernel_events is not None:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            if lattice_mode:
                ops.sdf_lattice(axes, scene.sv, sdf_w, u, x0, nx, sign=-1.0)
            else:
                xx, yy, zz = torch.meshgrid(axes[0][x0:x0 + nx], axes[1], axes[2], indexing="ij")
                pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).contiguous()
                sdf, _ = ops.sdf_mlp(pts, scene.sv, sdf_w, want_grad=False)
                u[x0:x0 + nx] = -sdf.view(nx, resolution, resolution)
            if self.kernel_events is not None:
                b.record()
                self.kernel_events.append(("sdf_grid", a, b))
        return u

    def extract_geometry(self, volumes, sparse_idxes, bound_min, bound_max, resolution, threshold, scene=None):
        """implicit_surface.py:337-357.  Callable with the reference's signature (volumes: (N_s, 7) rows fine -> coarse,
        sparse_idxes: their index tables) or with prepared SceneVolumes (`scene=`)."""
        from .marching_cubes import marching_cubes
        if scene is None:
            if volumes is None or sparse_idxes is None:
                raise ValueError("extract_geometry needs either (volumes, sparse_idxes) or scene=SceneVolumes")
            scene = _LatticeScene(volumes, sparse_idxes)
        u = self.sdf_grid(scene, bound_min, bound_max, resolution)
        b_max_np = bound_max.detach().cpu().numpy()
        b_min_np = bound_min.detach().cpu().numpy()
        # vertices / (resolution - 1.0) * (b_max - b_min) + b_min (:354-356) in float64 on the device, before the one copy to the host
        vertices, triangles = marching_cubes(u, threshold, rescale=(resolution - 1.0, (b_max_np - b_min_np).astype(np.float64),
                                                                     b_min_np.astype(np.float64)))
        return vertices, triangles

    def validate(self, rays_o, rays_d, near, far, scene, bound_min, bound_max, hw, cos_anneal_ratio=1.0, step=None,
                 extract_geometry=True, mesh_resolution=512, threshold=0.0, chunk=None):
        """implicit_surface.py:359-402.  The reference's 256-ray chunks exist to bound autograd memory; here rays are
        independent, so `chunk` is only a scratch-size knob: default `render.val_chunk` (conf key of ours, 2^19 rays: a 576 x 800
        image in one launch of every kernel - eight 65,536-ray chunks cost 6 ms more per image in launch tails and small
        launches), capped so that a chunk's per-sample buffers (~128 B per sample: points, masks, SDF + gradient, colours, the
        gradient kernel's scratch) take at most half of the device memory that is free right now - ranks that share a device
        (`--one-gpu`, scene-parallel runs on fewer GPUs than ranks) then shrink their chunks instead of running out of memory.
        With render.perturb > 0 the jitters are drawn in the
        reference's order (per 256-ray block, stages inner: draw_jitter), so a seeded run reproduces the reference's
        sample positions whatever `chunk` is."""
        outputs = {}
        height, width = int(hw[0]), int(hw[1])
        cols, nrms, sdeps, rdeps = [], [], [], []
        jitter = self.draw_jitter(rays_o.shape[0], ref_chunk=256) if self.perturb > 0 else None
        chunk = self.val_chunk_rays(rays_o.device) if chunk is None else int(chunk)
        for s in range(0, rays_o.shape[0], chunk):
            o = self.render_scene(rays_o[s:s + chunk], rays_d[s:s + chunk], near[s:s + chunk], far[s:s + chunk], scene,
                                  cos_anneal_ratio, per_sample=False, jitter=None if jitter is None else jitter[s:s + chunk])
            cols.append(o["color_fine"])
            nrms.append(o["normal_val"])
            sdeps.append(o["sdf_depth"])
            rdeps.append(o["render_depth"])
        # Round 5: the image post-processing of :377-399 (x 256 / rot . n x 128 + 128, clip) runs on the device and the five arrays
        # leave it as ONE pinned buffer on a side stream while the lattice kernels of extract_geometry run (the reference's order -
        # geometry first, then one .cpu() per output and numpy post-processing - serialises ~20 ms of copies and host passes
        # behind 270 ms of kernels); the host waits for the copy once, after the mesh is there.
        R = rays_o.shape[0]
        color, nrm = torch.cat(cols), torch.cat(nrms)
        rot = torch.from_numpy(np.ascontiguousarray(scene.cams.rot_ref)).to(color.device, torch.float32)
        img = (color * 256).clamp(0, 255)
        nimg = ((nrm @ rot.t()) * 128 + 128).clamp(0, 255)
        flat = torch.cat([color.reshape(-1), img.reshape(-1), nimg.reshape(-1), torch.cat(sdeps).reshape(-1), torch.cat(rdeps).reshape(-1)])
        host, copied = self._to_host_async(flat)
        if extract_geometry:
            v, t = self.extract_geometry(None, None, bound_min, bound_max, mesh_resolution, threshold, scene=scene)
            outputs["vertices"], outputs["triangles"] = v, t
        if copied is not None:
            copied.synchronize()
        outputs["color_fine"] = host[:3 * R].view(R, 3)
        outputs["img_fine"] = host[3 * R:6 * R].numpy().reshape([height, width, 3])
        outputs["normal_img"] = host[6 * R:9 * R].numpy().reshape([height, width, 3])
        outputs["sdf_depth"] = host[9 * R:10 * R].numpy().reshape([height, width])
        outputs["render_depth"] = host[10 * R:11 * R].numpy().reshape([height, width])
        return outputs

    def _to_host_async(self, t):
        """Device tensor -> (pinned host tensor, event): copied on a side stream that waits for the work queued so far; the caller
        synchronises on the event before reading.  The buffer comes from torch's caching host allocator and belongs to the
        result (views of it are what validate returns).  CPU tensors pass through (event None)."""
        if not t.is_cuda:
            return t, None
        buf = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        if getattr(self, "_copy_stream", None) is None or self._copy_stream.device != t.device:
            self._copy_stream = torch.cuda.Stream(device=t.device)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ready)
            buf.copy_(t, non_blocking=True)
            t.record_stream(self._copy_stream)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return buf, done

    def val_chunk_rays(self, device):
        """Rays per launch of `validate`: render.val_chunk (default 2^19), at most what half of the free device memory holds."""
        chunk = self.val_chunk
        if torch.device(device).type == "cuda":
            free, _ = torch.cuda.mem_get_info(device)
            per_ray = 128 * sum(self.n_samples) + 256
            fit = (free // 2) // per_ray
            chunk = int(max(256, min(chunk, (fit // 256) * 256)))
        return chunk

    def pseudo_sdf(self, pseudo_pts, scene):
        """implicit_surface.py:425-434: the SDF at the dataset's pseudo surface points (zero where no level is occupied), the
        input of `pseudo_sdf_loss` (losses/loss.py:67).  Recorded for `backward_render` when a training forward was."""
        pp = pseudo_pts.float().contiguous()
        occ = scene.occupied_any(pp)
        sdf_w, _ = self.packed_weights(pp.device)
        sdf, _ = ops.sdf_mlp(pp, scene.sv, sdf_w, mask=occ.to(torch.uint8), want_grad=False)
        if getattr(self, "_ctx", None) is not None:
            self._ctx["pseudo_pts"], self._ctx["pseudo_occ"] = pp, occ
            _occupied_list(self._ctx, "pseudo_occ")
        return torch.where(occ, sdf, torch.zeros_like(sdf))[:, None]

    def forward(self, mode, ipts, matching_volume, volumes, sparse_idxes, mask_volumes, features, match_features,
                cos_anneal_ratio=1.0, step=None):
        """implicit_surface.py:404-436 with the reference's argument list (volumes (N_s, 7) rows, int64 index tables and NCHW
        feature maps, all fine -> coarse), for a models/surf.py that swaps this class in.  In train mode
        with autograd enabled the outputs carry grad_fn and `volumes` / `features` receive their gradients
        (surf_amd.autograd.differentiable_render)."""
        rays_o, rays_d = ipts["rays_o"], ipts["rays_d"]
        near, far = ipts["near"], ipts["far"]
        if near.shape[0] == 1:
            near = near.repeat(rays_o.shape[0], 1)
            far = far.repeat(rays_o.shape[0], 1)
        with torch.no_grad():
            scene = self.scene(matching_volume, volumes, sparse_idxes, mask_volumes, features, ipts["imgs"], ipts["intrs"],
                               ipts["c2ws"])
            if match_features is not None and mode != "val":
                scene.match_feats_t4 = [ops.pack_texel4(f.detach().float().contiguous()) for f in match_features]
            if mode == "val":
                outputs = self.validate(rays_o, rays_d, near, far, scene, ipts["bound_min"], ipts["bound_max"], ipts["hw"],
                                        cos_anneal_ratio, step)
                if "pseudo_pts" in ipts:  # implicit_surface.py:425-434
                    outputs["pseudo_sdf"] = self.pseudo_sdf(ipts["pseudo_pts"], scene)
                return outputs

        def run():
            out = self.render_scene(rays_o, rays_d, near, far, scene, cos_anneal_ratio, patch_warp=True, step=step)
            if "pseudo_pts" in ipts:
                out["pseudo_sdf"] = self.pseudo_sdf(ipts["pseudo_pts"], scene)
            return out

        graph_inputs = list(self.parameters()) + list(volumes) + list(features)
        if torch.is_grad_enabled() and any(t.requires_grad for t in graph_inputs):
            from . import autograd
            return autograd.differentiable_render(self, run, volumes, features)
        with torch.no_grad():
            return run()
