"""Drop-in for models/surf.py SuRF: same constructor, parameter names, ``forward(mode, ipts, cos_anneal_ratio,
step)`` and output keys; every stage runs in the HIP kernels of libsurf_hip.so.

``mode == "val"`` (and any forward under ``torch.no_grad()``) is the inference path.  A train-mode forward with autograd
enabled returns outputs that carry ``grad_fn`` (``surf_amd.autograd``: two ``torch.autograd.Function`` nodes, the volume
build + FPN and the render, whose backward passes are the HIP backward kernels), so the reference's own loop -
``loss = Loss(outputs, inputs, step)["loss"]; loss.backward(); optimizer.step()`` (runner.py:155-165) - and
``DistributedDataParallel(model)`` (runner.py:102) work unchanged.  ``render.perturb > 0`` (the per-ray jitter of
``ImplicitSurface.render``, active in ``val`` too) is supported, and so is the per-scene volume API of finetuning
(``has_vol`` / ``init_volumes`` / ``get_params_vol`` / ``load_params_vol``, surf.py:47-78).
"""
import torch
import torch.nn as nn

from . import ops
from .grads import accumulate
from .feature_network import FeatureNetwork
from .implicit_surface import ImplicitSurface, SceneVolumes
from .matching_field import MatchingField
from .reg_network import SparseCostRegNetList
from .volume import Volume


class SuRF(nn.Module):
    def __init__(self, confs):
        super().__init__()
        self.has_vol = confs.get_bool("has_vol", default=False)
        self.range_ratios = [float(r) for r in confs.get_list("range_ratios")]
        self.num_stage = len(self.range_ratios)
        if self.num_stage != 4:
            raise NotImplementedError("the kernels are built for the 4-stage pyramid of confs/*.conf")
        if not self.has_vol:                       # surf.py:24-32: a has_vol model carries only the implicit surface
            self.feature_network = FeatureNetwork(confs["feature_network"])
            self.volume = Volume(confs["volume"])
            self.reg_network = SparseCostRegNetList(confs["reg_network"])
            self.matching_field = MatchingField(confs["matching_field"])
            self.match_feature_network = FeatureNetwork(confs["feature_network"])
            for p in self.match_feature_network.parameters():
                p.requires_grad = False
        self.implicit_surface = ImplicitSurface(confs["implicit_surface"])
        self._vol_scene = None                     # kernel-layout copy of the frozen volumes (per view subset)
        # optional key (ours): reduced-precision policy of the training backward, BASELINE configs[3] ("fp32" | "bf16", see
        # ops.set_train_precision: which tensors may be bf16); applied process-wide at every train-mode forward of this model
        self.train_precision = confs.get_string("train_precision", "fp32")
        if self.train_precision not in ops.TRAIN_PRECISIONS:
            raise ValueError(f"train_precision must be one of {sorted(ops.TRAIN_PRECISIONS)}, got {self.train_precision!r}")

    def get_optim_params(self, lr_conf):
        """surf.py:36-45 (parameter groups; training itself is row f2)."""
        groups = [{"params": list(self.implicit_surface.parameters()), "lr": lr_conf["mlp_lr"]}]
        if not self.has_vol:
            feat = list(self.feature_network.parameters()) + list(self.reg_network.parameters()) + list(self.volume.parameters())
            groups.append({"params": feat, "lr": lr_conf["feat_lr"]})
        else:
            for v_lr, volume_param in zip(lr_conf["vol_lr"], self.volumes):
                groups.append({"params": volume_param, "lr": v_lr})
        return groups

    # ---- per-scene volumes (finetuning API, surf.py:47-78) ----------------------------------------------------------
    @torch.no_grad()
    def init_volumes(self, ipts):
        """surf.py:65-78: run the FPN and the 4-stage build once and keep the results as (frozen-structure) parameters:
        `volumes` = the (N_s, 7) feature rows (trainable in the reference), `sparse_idxes` = the index tables,
        `matching_volume`, `features` = the FPN maps (nv, 4, h, w) coarse -> fine.  `mask_volmes` (the reference's dense
        0/1 volumes, 1.4 GB at 704^3) are implied by the tables (mask == table >= 0) and materialised only on demand."""
        feats_t4 = self.feature_network(ipts["imgs"])
        _, volumes, tables, mvol = self.build_volumes(ipts, feats_t4)
        self.volumes = nn.ParameterList([nn.Parameter(v[:, 1:].contiguous(), requires_grad=True) for v in volumes])
        self.sparse_idxes = nn.ParameterList([nn.Parameter(t, requires_grad=False) for t in tables])
        self.matching_volume = nn.Parameter(mvol[None, None], requires_grad=False)
        self.features = [f.permute(0, 3, 1, 2).contiguous() for f in feats_t4]
        self.has_vol = True
        self._vol_scene = None

    @property
    def mask_volmes(self):
        """The reference's dense mask volumes (1,1,D,D,D), built from the index tables when somebody asks for them."""
        return [(t.detach() >= 0).float()[None, None] for t in self.sparse_idxes]

    def get_params_vol(self):
        """surf.py:56-63, plus the two entries the reference forgets (`sparse_idxes`, `matching_volume`: without them
        its own load_params_vol cannot render)."""
        return {"volumes": self.volumes, "mask_volmes": self.mask_volmes, "features": self.features,
                "implicit_surface": self.implicit_surface.state_dict(),
                "sparse_idxes": [t.detach() for t in self.sparse_idxes], "matching_volume": self.matching_volume.detach()}

    def load_params_vol(self, path, device):
        """surf.py:47-54.  Accepts this class's get_params_vol files; a file written by the reference lacks the index
        tables and the matching volume (its loader has the same gap) and is refused with an explanation."""
        model = torch.load(path, map_location="cpu", weights_only=False)["model"]
        missing = [k for k in ("sparse_idxes", "matching_volume") if k not in model]
        if missing:
            raise KeyError(f"{path}: no {missing} in the saved volumes.  The reference's get_params_vol (surf.py:56-63) does not "
                           "save them and its load_params_vol cannot render either; re-save with surf_amd's get_params_vol")
        self.volumes = nn.ParameterList([nn.Parameter(v.detach().to(device).float(), requires_grad=True) for v in model["volumes"]])
        self.sparse_idxes = nn.ParameterList([nn.Parameter(t.to(device).to(torch.int32), requires_grad=False)
                                              for t in model["sparse_idxes"]])
        self.matching_volume = nn.Parameter(model["matching_volume"].to(device).float(), requires_grad=False)
        self.features = [f.to(device).float() for f in model["features"]]
        self.implicit_surface.load_state_dict(model["implicit_surface"])
        self.has_vol = True
        self._vol_scene = None

    def backward(self, g_color, g_depth=None, g_gradient_error=0.0, g_sparse_sdf=None, g_ncc=None, g_smooth_error=0.0,
                 g_pseudo_sdf=None):
        """Partial backward of the last train-mode forward (row f2): `.grad` of every implicit-surface parameter and, in
        finetune mode (has_vol), of the per-scene feature volumes - what surf.py:36-45 hands the optimiser there.  See
        ImplicitSurface.backward_render for the chain; the volume build / FPN backward is `backward_volumes`.  (The explicit
        form of what `loss.backward()` does through surf_amd.autograd.)"""
        gfeats = None
        if getattr(self, "_train_tape", None) is not None:       # volume-building model: the colour path's share of d FPN maps
            self._train_tape["gfeats"] = [torch.zeros_like(f) for f in self._train_tape["feats"]]      # coarse -> fine
            gfeats = self._train_tape["gfeats"][::-1]
        with ops.precision_scope(getattr(self, "_fwd_precision", None)):
            dvols = self.implicit_surface.backward_render(g_color, g_depth, g_gradient_error, g_sparse_sdf, g_ncc, gfeats_t4=gfeats,
                                                          g_smooth_error=g_smooth_error, g_pseudo_sdf=g_pseudo_sdf)
        if self.has_vol:
            for p, g in zip(self.volumes, dvols[::-1]):          # volumes are kept coarse -> fine
                accumulate(p, g)
        return dvols

    def _match_stage(self, t, s, g_depths, d_mvol):
        """Stage s's share of the matching chain of `_backward_volumes`: where a depth term reached this stage or a finer one,
        the matching-field backward into the dense volume gradient, then the densify backward into the rows' logit column and
        the coarser stage's volume gradient.  Returns (g_logit (N_s, 1) or None, d_mvol of stage s - 1 or None)."""
        r = t["vol"][s]
        cams, dev = t["cams"], t["feats"][0].device
        nv = t["feats"][0].shape[0]
        H, W = t["hw"]
        gd = (g_depths or {}).get(s, (None, None))
        if gd[0] is not None or gd[1] is not None:
            g_full = torch.zeros(nv, H, W, dtype=torch.float32, device=dev)
            if gd[0] is not None:
                g_full[0] += gd[0]
            if gd[1] is not None:
                g_full[t["src_idx"]] += gd[1]
            d_mvol = self.matching_field.backward(cams, t["near_fars"], (H, W), r["mvol"], s, self.range_ratios, g_full,
                                                  r["pre_depths"], r.get("jitter"), dmvol=d_mvol, src_idx=t["src_idx"],
                                                  stats=r.get("stats"))
        g_logit = d_prev = None
        if d_mvol is not None:
            g_logit = torch.zeros(r["coords"].shape[0], 1, dtype=torch.float32, device=dev)
            if s > 0:
                Dp = r["D"] // 2
                d_prev = torch.zeros(Dp, Dp, Dp, dtype=torch.float32, device=dev)
            ops.densify_backward(r["coords"], r["table"], d_mvol, g_logit, d_prev)
        return g_logit, d_prev

    @torch.no_grad()
    def start_matching_chain(self, t, g_depths, consumer=None):
        """The matching chain of the tape `t` for the depth gradients `g_depths` ({stage: (d depth_stage, d depth_src_stage)}) on
        its own stream (ops.SideStream lane 2; `consumer`: the stream that will read the results when the caller already runs on
        the lane), fine -> coarse.  Returns {stage: (g_logit or None, event)} for
        `_backward_volumes(match=...)`, or None when the side streams are off (the sweep then runs the stages in line)."""
        dev = t["feats"][0].device
        if not (ops.side.active("match") and dev.type == "cuda"):
            return None
        inline = ops.side.on_lane(2)     # called from a graph node that lives on the lane (autograd._DepthTap): already there
        main = consumer if (inline and consumer is not None) else torch.cuda.current_stream()
        out = {}

        def chain():
            dm = None
            for s in range(self.num_stage - 1, -1, -1):
                g_logit, dm = self._match_stage(t, s, g_depths, dm)
                if g_logit is not None:
                    g_logit.record_stream(main)          # allocated in the lane's pool, consumed (and freed) on the sweep's stream
                out[s] = (g_logit, ops.side.mark())

        if inline:
            chain()
        else:
            with ops.side.fork(lane=2):
                chain()
            ops.side.keep(*[g for pair in (g_depths or {}).values() for g in pair], lane=2)
        return out

    def backward_volumes(self, row_grads_f2c, g_depths=None, tape=None, gfeats=None, sink=None, match=None):
        """See _backward_volumes.  With the module's own tape (tape=None) it runs under the training-precision policy that forward
        was recorded with (a graph node passes its tape AND sets its own scope: surf_amd.autograd._Build)."""
        if tape is not None:
            return self._backward_volumes(row_grads_f2c, g_depths, tape, gfeats, sink, match)
        with ops.precision_scope(getattr(self, "_fwd_precision", None)):
            return self._backward_volumes(row_grads_f2c, g_depths, tape, gfeats, sink, match)

    def _backward_volumes(self, row_grads_f2c, g_depths=None, tape=None, gfeats=None, sink=None, match=None):
        """Backward of the volume build + FPN of the last `forward("train", ..., record=True)` (surf.py:80-131 under
        loss.backward()): row_grads_f2c = d loss / d the stages' feature rows, fine -> coarse, (N_s, 7) (what
        ImplicitSurface.backward_render returns) or (N_s, 8) = [logit | 7 features] rows, None = no gradient; g_depths =
        {stage: (d loss / d depth_stage{s} (H,W) or None, d loss / d depth_src_stage{s} or None)}.  Accumulates `.grad` of
        every parameter of reg_network, volume.agg_mlp and feature_network (or fills `sink`, a grads.GradSink).  Per stage,
        fine -> coarse: matching-field backward -> densify backward (-> the coarser matching volume) -> sparse U-Net backward
        -> cost-volume backward (-> FPN maps, agg_mlp) and the parent-feature scatter (-> the coarser stage's `mid` rows);
        then the FPN backward on the maps' total gradient (cost volumes + the colour path's share: `gfeats`, coarse -> fine,
        or what `SuRF.backward` left in the tape).  Not differentiable: the voxel selections, the detached depths.
        tape: the record to differentiate (default: the module's last one, consumed).  match: what start_matching_chain(tape,
        g_depths) returned if the caller has launched the matching chain already."""
        own = tape is None
        t = self._train_tape if own else tape
        if t is None:
            raise RuntimeError("backward_volumes needs forward('train', ..., record=True) of a volume-building model first")
        if not t.get("bn_train", self.reg_network.training):
            raise RuntimeError("backward_volumes: the sparse U-Net's tape is recorded in train mode (model.train())")
        feats, cams = t["feats"], t["cams"]
        dev = feats[0].device
        nv = feats[0].shape[0]
        H, W = t["hw"]
        if gfeats is None:                           # SuRF.backward leaves the colour path's share in the tape
            gfeats = t.pop("gfeats", None) or [torch.zeros_like(f) for f in feats]
        g_agg = torch.zeros(49, dtype=torch.float32, device=dev)
        n = self.num_stage
        d_mvol, d_mid = None, None
        pending = []

        # The matching chain (fine -> coarse: matching-field backward -> densify backward -> the coarser stage's volume gradient)
        # depends on the other chain (sparse U-Net -> cost volume -> the coarser stage's `mid` rows) nowhere: it only FEEDS it the
        # logit column of one g_out per stage.  matching_depth_bwd is bound by the probing of its LDS hash (5 ms a step on a
        # fraction of the chip's wave slots), so the whole chain runs ahead on its own stream (start_matching_chain) - from here,
        # or already from the graph's depth tap (surf_amd.autograd._DepthTap: before the render's backward), `match`.
        if match is None:
            match = self.start_matching_chain(t, g_depths)
        for s in range(n - 1, -1, -1):
            r = t["vol"][s]
            rg = row_grads_f2c[n - 1 - s]
            if rg is None:
                g_out = torch.zeros(r["coords"].shape[0], 8, dtype=torch.float32, device=dev)
            elif rg.shape[1] == 8:
                g_out = rg.float().clone()                                  # written into below: never the caller's tensor
            else:
                g_out = torch.cat([rg.new_zeros(rg.shape[0], 1), rg], dim=1).float()
            if match is not None:
                g_logit, ev = match[s]
                if ev is not None:
                    ops.side.wait_for(ev)
            else:
                g_logit, d_mvol = self._match_stage(t, s, g_depths, d_mvol)
            if g_logit is not None:
                g_out[:, 0] += g_logit.view(-1)
            d_reg_in = self.reg_network.nets[s].backward(r["reg_tape"], g_out, d_mid, sink=sink, pending=pending)
            d_mid = self.volume.stage_backward(s, r["D"], feats, gfeats, cams, r["coords"], d_reg_in, g_agg, r.get("pidx"),
                                               r["n_parents"])
        ops.side.join(lanes=(ops.COSTVOL_LANE,))     # the stages' cost-volume backward launches (gfeats, g_agg are complete now)
        self.volume.assign_agg_grad(g_agg, sink=sink)
        self.feature_network.backward(t["fpn"], gfeats, sink=sink)
        ops.side.join()                              # the U-Nets' kernel gradients ran on the side stream beside all of the above
        for finish in pending:
            finish()
        self.last_voxels_per_stage = [int(r["coords"].shape[0]) for r in t["vol"]]
        if own:
            self._train_tape = None                 # the tapes hold every stage's activations (GBs at full size): one backward each
        return gfeats

    def _frozen_scene(self, ipts):
        """SceneVolumes of the frozen volumes for the views of this call (surf.py:150-156: features[view_ids])."""
        view_ids = [int(v) for v in ipts["view_ids"]] if "view_ids" in ipts else list(range(ipts["imgs"].shape[0]))
        key = (tuple(view_ids), tuple((p._version, p.data_ptr()) for p in self.volumes))
        if self._vol_scene is None or self._vol_scene[0] != key:
            feats = [f[view_ids].contiguous() for f in self.features]
            scene = SceneVolumes(self.matching_volume, [v.detach() for v in self.volumes][::-1],
                                 [t.detach() for t in self.sparse_idxes][::-1], feats[::-1], ipts["imgs"], ipts["intrs"],
                                 ipts["c2ws"])
            self._vol_scene = (key, scene)
        return self._vol_scene[1]

    @torch.no_grad()
    def build_volumes(self, ipts, features_c2f, cams=None, logit_override=None, timings=None, trace=None, perturb=False,
                      tape=None):
        """surf.py:80-131 (perturb = the train-mode z jitter of the matching field, surf.py:139).  features_c2f: texel4 maps coarse -> fine.
        Returns (outputs, volumes, tables, matching_volume) with per-stage lists coarse -> fine.
        `logit_override(coords, D) -> (N,)` (bench / tests only) replaces the U-Net's matching logit so that an
        untrained network still produces a realistic, surface-concentrated pyramid; `timings` (dict) receives
        per-stage HIP-event pairs; `trace` (dict, tests only) receives every stage's intermediate tensors; `tape` (list,
        train mode) receives per stage what `backward_volumes` needs."""
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
            rec = {} if tape is not None else None
            coords, reg_in = self.volume.stage_inputs(s, D, features_c2f, cams, coords, mid, depths,
                                                      base_range * self.range_ratios[s], saved=rec)
            if ev: ev[1].record()
            table = ops.table_from_coords(coords, D)
            reg_tape = [] if tape is not None else None
            out, mid = self.reg_network(reg_in, coords, D, s, table=table, tape=reg_tape)
            if logit_override is not None:
                out[:, 0] = logit_override(coords, D)
            if ev: ev[2].record()
            mvol, table = ops.densify(coords, out, D, mvol)
            if ev: ev[3].record()
            depths = self.matching_field(cams, ipts["near_fars"], (H, W), mvol, s, self.range_ratios, depths,
                                         return_lr=trace is not None, perturb=perturb,
                                         src_idx=int(ipts["src_idx"]) if "src_idx" in ipts else 0, saved=rec)
            if tape is not None:
                rec.update(D=D, coords=coords, reg_tape=reg_tape, table=table, mvol=mvol, pre_depths=pre_depths,
                           n_parents=0 if parents is None else int(parents.shape[0]))
                tape.append(rec)
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
    def run_build(self, mode, ipts, record=False):
        """FPN + 4-stage volume build of one forward (surf.py:136-139).  Returns (depth outputs, volumes coarse -> fine
        (N_s, 8) = [logit | 7 features], index tables, matching volume, texel4 feature maps coarse -> fine, cameras, tape or
        None).  record: keep what `backward_volumes` needs."""
        imgs, intrs, c2ws = ipts["imgs"], ipts["intrs"], ipts["c2ws"]
        cams = ops._cams_ext(ops.Cameras(intrs, c2ws), intrs, c2ws)
        fpn_tape, vol_tape = ([], []) if record else (None, None)
        features = self.feature_network(imgs, tape=fpn_tape)                # texel4, coarse -> fine
        outputs, volumes, tables, mvol = self.build_volumes(ipts, features, cams, perturb=(mode == "train"), tape=vol_tape,
                                                            logit_override=getattr(self, "logit_override", None))  # surf.py:139
        tape = None
        if record:
            tape = dict(fpn=fpn_tape, vol=vol_tape, feats=features, cams=cams, near_fars=ipts["near_fars"],
                        hw=tuple(imgs.shape[-2:]), src_idx=int(ipts["src_idx"]) if "src_idx" in ipts else 0,
                        bn_train=self.reg_network.training)
        return outputs, volumes, tables, mvol, features, cams, tape

    @torch.no_grad()
    def start_match_features(self, mode, ipts, step=None):
        """The frozen matching FPN's maps of a training forward (surf.py:141-148: loss-only inputs, read by the patch warp at the
        end of the render).  They depend on the images alone, so the forward launches them FIRST, on their own stream beside the
        whole volume build (ops.SideStream lane 3).  Returns (maps fine -> coarse, event or None) or None in `val`."""
        if mode == "val":
            return None
        imgs = ipts["imgs"]
        if step is not None and step % 2 == 0:                              # refresh the frozen matching FPN
            self.match_feature_network.load_state_dict(self.feature_network.state_dict(), strict=True)
            for p in self.match_feature_network.parameters():
                p.requires_grad = False
        if ops.side.active("fwd") and imgs.is_cuda:
            main = torch.cuda.current_stream()
            with ops.side.fork(lane=3):
                feats = self.match_feature_network(imgs)[::-1]
                ev = ops.side.mark()
            for f in feats:
                f.record_stream(main)
            return feats, ev
        return self.match_feature_network(imgs)[::-1], None

    @torch.no_grad()
    def build_scene(self, mode, ipts, volumes, tables, mvol, features, cams, step=None, match=None):
        """SceneVolumes of a freshly built pyramid (kernel layouts, no re-packing) + the frozen matching FPN's maps of a
        training forward (`match`: what start_match_features returned before the build; None: computed here)."""
        imgs = ipts["imgs"]
        scene = SceneVolumes.from_device_layouts(mvol, [v[:, 1:] for v in volumes[::-1]], tables[::-1], features[::-1],
                                                 ops.pack_texel4(imgs.detach().float().contiguous()), cams)
        if mode != "val":
            if match is None:
                match = self.start_match_features(mode, ipts, step)
            scene.match_feats_t4, scene.match_ready = match
        return scene

    @torch.no_grad()
    def run_render(self, mode, ipts, scene, cos_anneal_ratio=1.0, step=None):
        """ImplicitSurface.forward on prepared SceneVolumes (surf.py:159)."""
        isurf = self.implicit_surface
        rays_o, rays_d = ipts["rays_o"], ipts["rays_d"]
        near, far = ipts["near"], ipts["far"]
        if near.shape[0] == 1:
            near = near.repeat(rays_o.shape[0], 1)
            far = far.repeat(rays_o.shape[0], 1)
        if mode == "val":
            return isurf.validate(rays_o, rays_d, near, far, scene, ipts["bound_min"], ipts["bound_max"], ipts["hw"],
                                  cos_anneal_ratio, step, mesh_resolution=int(ipts.get("mesh_resolution", 512)))
        surface = isurf.render_scene(rays_o, rays_d, near, far, scene, cos_anneal_ratio, patch_warp=True, step=step)
        if "pseudo_pts" in ipts:                                            # implicit_surface.py:425-434
            surface["pseudo_sdf"] = isurf.pseudo_sdf(ipts["pseudo_pts"], scene)
        return surface

    def _wants_graph(self, mode, record=False):
        """A train-mode forward records an autograd graph when autograd is on and something is trainable.  A volume-building
        model must also be in train() mode, as runner.py:143 puts it: the sparse U-Net's backward kernels are those of
        batch-statistics BatchNorm (an eval()-mode model forwards without a graph)."""
        if mode == "val" or record or not torch.is_grad_enabled():
            return False
        if not self.has_vol and not self.reg_network.training:
            return False
        return any(p.requires_grad for p in self.parameters())

    def forward(self, mode, ipts, cos_anneal_ratio=1.0, step=None, record=False):
        """surf.py:133-163.  With autograd enabled a train-mode forward is differentiable (surf_amd.autograd): its outputs
        carry grad_fn and `loss.backward()` runs the HIP backward kernels.  record=True (surf_amd.training's explicit step)
        instead keeps the tapes on the module for `SuRF.backward` / `SuRF.backward_volumes` and returns plain tensors."""
        if mode != "val":
            ops.set_train_precision(self.train_precision)
            self._fwd_precision = ops.colgram_precision          # what the explicit backward entry points below run under
        if self._wants_graph(mode, record):
            from . import autograd
            return autograd.differentiable_forward(self, mode, ipts, cos_anneal_ratio, step)
        with torch.no_grad():
            self._train_tape = None
            if self.has_vol:                                                    # surf.py:149-156
                outputs, scene = {}, self._frozen_scene(ipts)
            else:
                match = self.start_match_features(mode, ipts, step)
                outputs, volumes, tables, mvol, features, cams, tape = self.run_build(mode, ipts, record=record)
                self._train_tape = tape
                scene = self.build_scene(mode, ipts, volumes, tables, mvol, features, cams, step, match=match)
            outputs.update(self.run_render(mode, ipts, scene, cos_anneal_ratio, step))
            return outputs
