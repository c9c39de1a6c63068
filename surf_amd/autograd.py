"""torch.autograd layer over the HIP forward / backward kernel pairs: what makes `loss.backward()` work.

The reference trains with plain autograd (runner.py:152-165):

    outputs = self.model(self.mode, inputs, cos_anneal_ratio=..., step=...)     # DistributedDataParallel(SuRF), runner.py:102
    loss = self.loss(outputs, inputs, step)["loss"]
    self.optimizer.zero_grad(); loss.backward(); self.optimizer.step()

The kernels keep no graph, so a train-mode `SuRF.forward` with autograd enabled is recorded as TWO graph nodes whose
backward passes are the HIP backward kernels:

  `_Build`  (volume-building models)  parameters of feature_network / volume.agg_mlp / reg_network
            -> per-stage rows [logit | 7 features], FPN maps (texel4), depth_stage{s}, depth_src_stage{s}
            backward = SuRF.backward_volumes (matching field -> densify -> sparse U-Net -> cost volume -> FPN)
  `_Render` parameters of implicit_surface (+ the per-stage rows, + the FPN maps)
            -> color_fine, render_depth, gradient_error, sparse_sdf, smooth_error, ref_gray_val, sampled_gray_val, pseudo_sdf
            backward = ImplicitSurface.backward_render (composite -> SDF / blend backward, patch tangents)

Autograd's engine sums the two consumers of the FPN maps (cost volumes and the colour path) and of the rows, runs
`_Render.backward` first and `_Build.backward` once every depth / row / map gradient has arrived; the parameter gradients
are RETURNED (grads.GradSink), so AccumulateGrad - and DistributedDataParallel's bucket hooks behind it - see them: the
implicit-surface bucket is all-reduced over RCCL while the volume-build backward still runs.  Everything that the
reference detaches (voxel selections, z-sampling from the matching volume, normals and feature maps of the patch warp,
`mid_inside_sphere`, `valid_mask`) is returned as plain tensors.

`lncc` / `photometric_loss` give the two HIP loss terms of `surf_amd.losses.Loss` a backward as well; a caller that keeps
the reference's own `Loss` (torch ops on the outputs) needs nothing else.
"""
import torch

from . import ops
from .grads import GradSink

RENDER_KEYS = ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "smooth_error", "ref_gray_val", "sampled_gray_val")


def _scalar(g):
    """Upstream gradient of a scalar output: a device scalar stays on the device (backward_render multiplies it in inside its
    kernels / tensor expressions: reading it back would stall the host at the head of the sweep); else a python float."""
    if g is None:
        return 0.0
    return g.detach() if torch.is_tensor(g) and g.is_cuda else float(g)


class _Render(torch.autograd.Function):
    @staticmethod
    def forward(ctx, isurf, cfg, *tensors):
        """tensors = implicit-surface parameters (cfg["n_params"]) + sparse rows fine -> coarse (cfg["n_rows"], (N_s, 7 or 8))
        + FPN maps fine -> coarse (the rest, may be none; cfg["feat_layout"]: "t4" texel4 (nv,h,w,4) or "nchw"): graph inputs
        only - cfg["run"]() renders from the prepared scene and returns (outputs dict, record of the forward); the
        non-differentiable outputs are handed back through cfg["holder"]."""
        ctx.set_materialize_grads(False)
        out, rec = cfg["run"]()
        n_params, n_rows = cfg["n_params"], cfg["n_rows"]
        ctx.isurf, ctx.rec = isurf, rec
        ctx.n_params, ctx.n_rows, ctx.n_feats = n_params, n_rows, len(tensors) - n_params - n_rows
        ctx.params = tensors[:n_params]
        ctx.row_widths = [int(t.shape[1]) for t in tensors[n_params:n_params + n_rows]]
        ctx.feat_layout = cfg.get("feat_layout", "t4")
        keys = RENDER_KEYS + (("pseudo_sdf",) if "pseudo_sdf" in out else ())
        ctx.keys = keys
        ctx.precision = ops.colgram_precision          # the policy this forward ran under is the one its backward runs under
        cfg["holder"].update({k: v for k, v in out.items() if k not in keys})
        cfg["holder"]["_keys"] = keys
        return tuple(out[k] for k in keys)

    @staticmethod
    def backward(ctx, *gouts):
        g = dict(zip(ctx.keys, gouts))
        isurf, rec = ctx.isurf, ctx.rec
        if rec is None:
            raise RuntimeError("the render of this forward was already differentiated (its record is freed after one backward)")
        sink = GradSink()
        scene = rec["scene"]
        first_feat = 2 + ctx.n_params + ctx.n_rows
        want_feats = ctx.n_feats > 0 and any(ctx.needs_input_grad[first_feat:])
        gfeats = [torch.zeros_like(f) for f in scene.feats_t4] if want_feats else None
        patches = (g["ref_gray_val"], g["sampled_gray_val"])
        with ops.precision_scope(ctx.precision):
            dvols = isurf.backward_render(g["color_fine"], g["render_depth"], _scalar(g["gradient_error"]), g["sparse_sdf"], None,
                                          gfeats_t4=gfeats, g_smooth_error=_scalar(g["smooth_error"]), g_pseudo_sdf=g.get("pseudo_sdf"),
                                          g_patches=None if patches == (None, None) else patches, ctx=rec, sink=sink, rows8=True)
        ctx.rec = None
        rows = []
        for dv, w in zip(dvols, ctx.row_widths):   # the kernels' [7 features | 0] rows -> the input's own width
            # [logit | 7] rows (logit gradient 0): the zero column moves to the front - one pass instead of slice + fill + copy
            rows.append(torch.roll(dv, 1, dims=1) if w == 8 else dv[:, :7].contiguous())
        feats = [None] * ctx.n_feats
        if want_feats:
            feats = [gf.permute(0, 3, 1, 2)[:, :4].contiguous() if ctx.feat_layout == "nchw" else gf for gf in gfeats]
        return (None, None) + tuple(sink.get(p) for p in ctx.params) + tuple(rows) + tuple(feats)


class _Build(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, mode, ipts, holder, *params):
        ctx.set_materialize_grads(False)
        outputs, volumes, tables, mvol, features, cams, tape = model.run_build(mode, ipts, record=True)
        ctx.model, ctx.tape, ctx.params = model, tape, params
        ctx.precision = ops.colgram_precision
        n = model.num_stage
        ctx.n = n
        holder.update(tables=tables, mvol=mvol, cams=cams, tape=tape)
        if features and features[0].is_cuda:
            holder["stream"] = torch.cuda.current_stream()       # where this node's backward will run (autograd's stream rule)
        ctx.holder = holder
        depths = [outputs[f"depth_stage{s}"] for s in range(n)] + [outputs[f"depth_src_stage{s}"][...] for s in range(n)]
        return tuple(volumes) + tuple(features) + tuple(depths)

    @staticmethod
    def backward(ctx, *g):
        n, model, tape = ctx.n, ctx.model, ctx.tape
        if tape is None:
            raise RuntimeError("the volume build of this forward was already differentiated (its tapes are freed after one backward)")
        g_rows, g_feats, g_dep, g_src = g[:n], g[n:2 * n], g[2 * n:3 * n], g[3 * n:4 * n]
        sink = GradSink()
        gfeats = [torch.zeros_like(f) if gf is None else gf.contiguous().clone() for f, gf in zip(tape["feats"], g_feats)]
        match = ctx.holder.pop("match", None)         # the matching chain, if the depth tap has launched it already
        ctx.holder["tape"] = None
        with ops.precision_scope(ctx.precision):
            model.backward_volumes(list(g_rows[::-1]), {s: (g_dep[s], g_src[s]) for s in range(n)}, tape=tape, gfeats=gfeats, sink=sink,
                                   match=match)
        ctx.tape = None
        return (None, None, None, None) + tuple(sink.get(p) for p in ctx.params)


class _DepthTap(torch.autograd.Function):
    """Identity on the 2 n depth maps of the volume build, placed in the graph AFTER the render node: autograd runs the younger
    node first, and the depth terms are the last ones the loss computes (losses/loss.py:47-63), so this backward sees the maps'
    gradients before the render's backward has started.  It launches the matching chain of the volume backward - which needs
    nothing but these gradients - on its own stream (SuRF.start_matching_chain), where it runs beside the whole render
    backward instead of in front of the sparse U-Nets' (-2.8 ms of exposed matching-field backward per step); `_Build.backward`
    picks the result up from `holder`.  Without side streams (or a loss without depth terms) it is a plain identity."""

    @staticmethod
    def forward(ctx, model, holder, *depths):
        ctx.set_materialize_grads(False)
        ctx.model, ctx.holder = model, holder
        return tuple(d.view_as(d) for d in depths)

    @staticmethod
    def backward(ctx, *g):
        holder = ctx.holder
        tape = holder.get("tape")
        if tape is not None and "match" not in holder and any(x is not None for x in g):
            n = len(g) // 2
            match = ctx.model.start_matching_chain(tape, {s: (g[s], g[n + s]) for s in range(n)}, consumer=holder.get("stream"))
            if match is not None:
                holder["match"] = match
        return (None, None) + tuple(g)


def differentiable_forward(model, mode, ipts, cos_anneal_ratio=1.0, step=None):
    """`SuRF.forward` in train mode with a graph (see the module docstring).  Same output dictionary as the plain path."""
    isurf = model.implicit_surface
    outputs = {}
    if model.has_vol:                                                   # surf.py:149-156: rows = the per-scene parameters
        scene = model._frozen_scene(ipts)
        rows = list(model.volumes)[::-1]                                # fine -> coarse
        feats = []
    else:
        holder = {}
        bparams = [p for m in (model.feature_network, model.volume, model.reg_network) for p in m.parameters() if p.requires_grad]
        with torch.no_grad():
            match = model.start_match_features(mode, ipts, step)           # on its own stream, beside the volume build
        res = _Build.apply(model, mode, ipts, holder, *bparams)
        n = model.num_stage
        volumes, features = list(res[:n]), list(res[n:2 * n])
        for s in range(n):
            outputs[f"depth_stage{s}"] = res[2 * n + s]
            outputs[f"depth_src_stage{s}"] = res[3 * n + s]
        with torch.no_grad():
            scene = model.build_scene(mode, ipts, [v.detach() for v in volumes], holder["tables"], holder["mvol"],
                                      [f.detach() for f in features], holder["cams"], step, match=match)
        rows, feats = volumes[::-1], features[::-1]
    outputs.update(_render_node(isurf, lambda: model.run_render(mode, ipts, scene, cos_anneal_ratio, step), rows, feats, "t4"))
    if not model.has_vol:
        n = model.num_stage
        keys = [f"depth_stage{s}" for s in range(n)] + [f"depth_src_stage{s}" for s in range(n)]
        depths = [outputs[k] for k in keys]
        if ops.side.active("match") and ops.lane_nodes and depths[0].is_cuda:
            # the tap LIVES on the matching lane: autograd runs a node's backward on the stream of its forward and orders the
            # gradients that cross streams, so the tap's backward - and the matching chain it launches - need no fork of their own,
            # and whatever feeds it from another lane (the photometric terms, surf_amd.losses) never touches the main stream
            lane = ops.side.lane_stream(2, depths[0].device)
            lane.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(lane):
                taps = _DepthTap.apply(model, holder, *depths)           # views: no kernel runs here
        else:
            taps = _DepthTap.apply(model, holder, *depths)
        outputs.update(zip(keys, taps))
    return outputs


def _render_node(isurf, run_render, rows, feats, feat_layout):
    rparams = [p for p in isurf.parameters() if p.requires_grad]
    holder = {}

    def run():
        out = run_render()
        rec, isurf._ctx = isurf._ctx, None          # the node owns the record of this forward (freed after its backward)
        return out, rec

    cfg = dict(run=run, holder=holder, n_params=len(rparams), n_rows=len(rows), feat_layout=feat_layout)
    res = _Render.apply(isurf, cfg, *rparams, *rows, *feats)
    outputs = {k: v for k, v in holder.items() if k != "_keys"}
    outputs.update(dict(zip(holder["_keys"], res)))
    return outputs


def differentiable_render(isurf, run_render, volumes, features):
    """`ImplicitSurface.forward` in train mode with a graph, for a caller that owns the volume build (the
    reference's models/surf.py with this ImplicitSurface swapped in): `volumes` = the reference's (N_s, 7) rows fine ->
    coarse and `features` = its NCHW FPN maps fine -> coarse, both with their torch autograd history; their gradients flow
    back into the reference's own modules.  run_render() -> the plain outputs of the train-mode render."""
    return _render_node(isurf, run_render, list(volumes), list(features), "nchw")


class _Lncc(torch.autograd.Function):
    """compute_LNCC2 (losses/ncc.py:7-51) = surf_lncc, backward surf_lncc_backward."""

    @staticmethod
    def forward(ctx, ref, src):
        ref, src = ref.float().contiguous(), src.float().contiguous()
        ctx.save_for_backward(ref, src)
        return ops.lncc(ref, src)

    @staticmethod
    def backward(ctx, g):
        ref, src = ctx.saved_tensors
        g_ref, g_src = ops.lncc_backward(ref, src, g)
        return g_ref, g_src


def lncc(ref_gray_val, sampled_gray_val):
    return _Lncc.apply(ref_gray_val, sampled_gray_val)


class _Photometric(torch.autograd.Function):
    """compute_ptloss (losses/photometric_loss.py:54-125) of one depth map = surf_ptloss_terms, backward surf_ptloss_backward."""

    @staticmethod
    def forward(ctx, depth, imgs_t4, mask, cams, ref_idx, topk):
        depth = depth.float().contiguous()
        loss, (warp, sums) = ops.photometric_loss(depth, imgs_t4, mask, cams, ref_idx=ref_idx, topk=topk, return_state=True)
        ctx.save_for_backward(depth, imgs_t4, mask, warp, sums)      # the warped images (29 MB a call at 5 x 576 x 800) are kept
        ctx.cams, ctx.ref_idx, ctx.topk = cams, ref_idx, topk         # rather than produced again by a second forward launch
        return loss

    @staticmethod
    def backward(ctx, g):
        depth, imgs_t4, mask, warp, sums = ctx.saved_tensors
        return ops.photometric_loss_backward(depth, imgs_t4, mask, ctx.cams, ctx.ref_idx, ctx.topk, upstream=g,
                                             state=(warp, sums)), None, None, None, None, None


def photometric_loss(depth, imgs_t4, mask, cams, ref_idx=0, topk=2):
    return _Photometric.apply(depth, imgs_t4, mask, cams, ref_idx, topk)


class _PhotometricMulti(torch.autograd.Function):
    """compute_ptloss of SEVERAL depth maps (the 2 n per-stage maps of losses/loss.py:47-55) as one graph node: the terms are
    independent of one another and each launch is bound by L2 reads and channel-planar atomics on a fraction of the chip, so
    both directions CAN deal them out over side streams (ops.SideStream lanes 4..7, user "loss": measured slower by 1.5 ms a
    step - the launches are bound by the memory system - and off by default; in line the node still saves autograd 2 n - 1
    node visits each way)."""
    LANES = (4, 5, 6, 7)

    @staticmethod
    def forward(ctx, imgs_t4, cams, specs, *depths):
        ctx.set_materialize_grads(False)
        depths = [d.float().contiguous() for d in depths]
        on_side = ops.side.active("loss") and imgs_t4.is_cuda
        res = []
        for i, (d, (mask, ref_idx, topk)) in enumerate(zip(depths, specs)):
            fn = lambda d=d, mask=mask, ref_idx=ref_idx, topk=topk: ops.photometric_loss(d, imgs_t4, mask, cams, ref_idx=ref_idx,   # noqa: E731
                                                                                      topk=topk, return_state=True)
            res.append(ops.side.run(fn, lane=_PhotometricMulti.LANES[i % 4], keep=(d,)) if on_side else fn())
        if on_side:
            ops.side.join(lanes=_PhotometricMulti.LANES)
        n = len(depths)
        masks = [m for m, _, _ in specs]
        # tensors through save_for_backward (autograd's in-place modification check covers them), the rest on ctx
        ctx.save_for_backward(imgs_t4, *depths, *masks, *[st[0] for _, st in res], *[st[1] for _, st in res])
        ctx.n, ctx.cams, ctx.views = n, cams, [(r, k) for _, r, k in specs]
        return tuple(loss for loss, _ in res)

    @staticmethod
    def backward(ctx, *gs):
        saved, n = ctx.saved_tensors, ctx.n
        imgs_t4 = saved[0]
        depths, masks, warps, sums = (saved[1 + k * n:1 + (k + 1) * n] for k in range(4))
        on_side = ops.side.active("loss") and imgs_t4.is_cuda
        out = []
        for i, (g, d, mask, (ref_idx, topk)) in enumerate(zip(gs, depths, masks, ctx.views)):
            if g is None:
                out.append(None)
                continue
            fn = lambda g=g, d=d, mask=mask, st=(warps[i], sums[i]), ref_idx=ref_idx, topk=topk: ops.photometric_loss_backward(  # noqa: E731
                d, imgs_t4, mask, ctx.cams, ref_idx, topk, upstream=g, state=st)
            out.append(ops.side.run(fn, lane=_PhotometricMulti.LANES[i % 4], keep=(g,)) if on_side else fn())
        if on_side:
            ops.side.join(lanes=_PhotometricMulti.LANES)
        return (None, None, None) + tuple(out)


def photometric_losses(depths, imgs_t4, cams, specs):
    """[compute_ptloss(depth_i) for i] with specs[i] = (mask (H,W) fp32, ref_idx, topk): one graph node, the launches side by side."""
    return _PhotometricMulti.apply(imgs_t4, cams, list(specs), *depths)


class _MaskedL1(torch.autograd.Function):
    """sum(|pred - target| mask) / (sum(mask) + 1e-8) (losses/loss.py:71-93) = surf_masked_l1, backward surf_masked_l1_backward:
    two launches instead of ~13 small torch kernels per term."""

    @staticmethod
    def forward(ctx, pred, target, mask):
        pred_c = pred.float().contiguous()
        target_c = target.detach().float().contiguous()
        out2 = ops.masked_l1(pred_c, target_c, mask)
        keep = (pred_c, target_c, out2) + (() if isinstance(mask, str) else (mask,))
        ctx.save_for_backward(*keep)
        ctx.mask_str = mask if isinstance(mask, str) else None
        ctx.shape = pred.shape
        return out2[0]

    @staticmethod
    def backward(ctx, g):
        saved = ctx.saved_tensors
        pred_c, target_c, out2 = saved[:3]
        mask = ctx.mask_str if ctx.mask_str is not None else saved[3]
        return ops.masked_l1_backward(pred_c, target_c, mask, out2, g).view(ctx.shape), None, None


def masked_l1(pred, target, mask):
    """mask: a float / bool tensor of pred's size, or the string "target>0"."""
    return _MaskedL1.apply(pred, target, mask)
