"""CPU: the autograd layer (surf_amd.autograd) with the HIP forward / backward routines replaced by small fakes of the same
shapes - what is tested is the GRAPH: that `loss.backward()` on the outputs of a train-mode `SuRF.forward` calls the render
backward first and the volume-build backward once all its upstream gradients arrived, hands every upstream gradient to the
right argument, returns one gradient per parameter (zeros where the sweep reached nothing) and that
`DistributedDataParallel(model)` (runner.py:102) averages them over two gloo ranks and starts from rank 0's parameters."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAKE = textwrap.dedent("""
    import torch
    from surf_amd import conf
    from surf_amd.grads import accumulate
    from surf_amd.surf import SuRF
    from tests.golden.make_golden import MODEL_CONF

    R, S, NV, H, W = 6, 5, 3, 8, 8
    N = [4, 6, 8, 10]                                   # voxels per stage, coarse -> fine

    def make_model(seed):
        cfg = dict(MODEL_CONF)
        cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
        torch.manual_seed(seed)
        model = SuRF(conf.from_dict(cfg)).train()
        calls = []

        def run_build(mode, ipts, record=False):
            calls.append("build_fwd")
            outs = {}
            for s in range(4):
                outs[f"depth_stage{s}"] = torch.full((H, W), 1.0 + s)
                outs[f"depth_src_stage{s}"] = torch.full((H, W), 2.0 + s)
            vols = [torch.ones(n, 8) for n in N]
            feats = [torch.ones(NV, H >> (3 - i), W >> (3 - i), 4) for i in range(4)]
            tape = {"feats": feats} if record else None
            return outs, vols, [None] * 4, None, feats, None, tape

        def build_scene(mode, ipts, volumes, tables, mvol, features, cams, step=None, match=None):
            return "scene"

        def run_render(mode, ipts, scene, cos_anneal_ratio=1.0, step=None):
            calls.append("render_fwd")
            model.implicit_surface._ctx = {"scene": type("S", (), {"feats_t4": [torch.ones(NV, H >> i, W >> i, 4) for i in range(4)]})()}
            out = {"color_fine": torch.ones(R, 3), "render_depth": torch.ones(R), "gradient_error": torch.tensor(0.5),
                   "sparse_sdf": torch.ones(1024 + R * S, 1), "smooth_error": torch.tensor(0.25),
                   "ref_gray_val": torch.ones(1, R, 121, 12), "sampled_gray_val": torch.ones(NV - 1, R, 121, 12),
                   "valid_mask": torch.ones(R, 1, dtype=torch.bool), "mid_inside_sphere": torch.ones(R, 1)}
            if "pseudo_pts" in ipts:
                out["pseudo_sdf"] = torch.ones(ipts["pseudo_pts"].shape[0], 1)
            return out

        seen = {}

        def backward_render(g_color, g_depth=None, g_gradient_error=0.0, g_sparse_sdf=None, g_ncc=None, gfeats_t4=None,
                            g_smooth_error=0.0, g_pseudo_sdf=None, g_patches=None, ctx=None, sink=None, rows8=False):
            calls.append("render_bwd")
            seen.update(g_color=g_color, g_depth=g_depth, g_eik=g_gradient_error, g_sparse=g_sparse_sdf, g_smooth=g_smooth_error,
                        g_pseudo=g_pseudo_sdf, g_patches=g_patches, ctx=ctx)
            k = float(g_color.sum())
            for i, p in enumerate(model.implicit_surface.parameters()):
                if i % 5 != 4:                              # every fifth parameter: the sweep reaches nothing -> zeros
                    accumulate(p, torch.full_like(p, k), sink)
            for i, gf in enumerate(gfeats_t4):
                gf += 10.0 * (i + 1)
            if rows8:       # the kernels' own rows: [7 features | 0] (the graph node asks for these and moves the zero column)
                return [torch.cat([torch.full((n, 7), 3.0), torch.zeros(n, 1)], dim=1) for n in N[::-1]]
            return [torch.full((n, 7), 3.0) for n in N[::-1]]

        def backward_volumes(row_grads_f2c, g_depths=None, tape=None, gfeats=None, sink=None, match=None):
            calls.append("build_bwd")
            seen.update(rows=row_grads_f2c, g_depths=g_depths, gfeats=gfeats, tape=tape)
            k = sum(float(g[0].sum()) for g in g_depths.values() if g[0] is not None)
            for m in (model.feature_network, model.volume, model.reg_network):
                for p in m.parameters():
                    accumulate(p, torch.full_like(p, k), sink)
            return gfeats

        model.run_build, model.build_scene, model.run_render = run_build, build_scene, run_render
        model.start_match_features = lambda mode, ipts, step=None: None        # the frozen matching FPN's launch (kernels)
        model.backward_volumes = backward_volumes
        model.implicit_surface.backward_render = backward_render
        return model, calls, seen

    def loss_of(out, scale=1.0):
        return scale * (2.0 * out["color_fine"].sum() + 0.1 * out["gradient_error"] + out["depth_stage2"].sum()
                        + (out["ref_gray_val"] * out["sampled_gray_val"]).sum() * 1e-3 + out["pseudo_sdf"].sum())
""")


def _exec_fake():
    ns = {}
    exec(FAKE, ns)
    return ns


def test_loss_backward_drives_both_nodes_and_fills_every_grad():
    ns = _exec_fake()
    model, calls, seen = ns["make_model"](0)
    ipts = {"pseudo_pts": torch.zeros(7, 3)}
    out = model("train", ipts, cos_anneal_ratio=0.7, step=3)
    for k in ("color_fine", "render_depth", "gradient_error", "sparse_sdf", "smooth_error", "ref_gray_val", "sampled_gray_val",
              "pseudo_sdf", "depth_stage0", "depth_src_stage3"):
        assert out[k].grad_fn is not None, k
    assert out["valid_mask"].grad_fn is None and out["valid_mask"].dtype == torch.bool      # detached in the reference too
    assert model.implicit_surface._ctx is None                                              # the node owns the record
    ns["loss_of"](out).backward()
    assert calls == ["build_fwd", "render_fwd", "render_bwd", "build_bwd"]
    R = ns["R"]
    assert torch.equal(seen["g_color"], torch.full((R, 3), 2.0)) and seen["g_depth"] is None
    assert abs(seen["g_eik"] - 0.1) < 1e-7 and seen["g_smooth"] == 0.0 and seen["g_sparse"] is None
    assert torch.equal(seen["g_pseudo"], torch.ones(7, 1))
    assert seen["g_patches"][0].shape == (1, R, 121, 12) and abs(float(seen["g_patches"][1].max()) - 1e-3) < 1e-9
    # the volume build received: the render's row gradients widened to [logit | 7] rows, the colour path's share of the FPN
    # maps (coarse -> fine), and only the depth map the loss used
    assert [tuple(r.shape) for r in seen["rows"]] == [(n, 8) for n in ns["N"][::-1]]
    assert all(float(r[:, 0].abs().max()) == 0.0 and float(r[:, 1:].min()) == 3.0 for r in seen["rows"])
    assert [float(g.flatten()[0]) for g in seen["gfeats"]] == [40.0, 30.0, 20.0, 10.0]
    assert seen["g_depths"][2][0] is not None and seen["g_depths"][2][1] is None and seen["g_depths"][0] == (None, None)
    for i, p in enumerate(model.implicit_surface.parameters()):
        assert p.grad is not None and float(p.grad.flatten()[0]) == (0.0 if i % 5 == 4 else 2.0 * R * 3)
    for name, p in model.named_parameters():
        if name.startswith("match_feature_network"):
            assert p.grad is None and not p.requires_grad
        elif not name.startswith("implicit_surface"):
            assert float(p.grad.flatten()[0]) == 64.0, name
    # one backward per forward: the records are gone
    out2 = model("train", ipts, 0.7, 3)
    ns["loss_of"](out2).backward(retain_graph=True)
    try:
        ns["loss_of"](out2).backward()
        raise AssertionError("a second backward through freed tapes must fail loudly")
    except RuntimeError as e:
        assert "already differentiated" in str(e)


def test_no_graph_under_no_grad_val_or_eval():
    ns = _exec_fake()
    model, calls, _ = ns["make_model"](0)
    with torch.no_grad():
        out = model("train", {"pseudo_pts": torch.zeros(2, 3)}, 1.0, 3)
    assert out["color_fine"].grad_fn is None
    model.eval()
    out = model("train", {"pseudo_pts": torch.zeros(2, 3)}, 1.0, 3)
    assert out["color_fine"].grad_fn is None
    assert not model._wants_graph("val") and not model.train()._wants_graph("train", record=True)


DDP_WORKER = FAKE + textwrap.dedent("""
    import json, os, sys
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    model, calls, seen = make_model(100 + rank)             # different seeds: DDP must start every rank from rank 0's state
    first = next(model.implicit_surface.parameters())
    before = float(first.detach().abs().sum())
    ddp = DistributedDataParallel(model)                    # runner.py:102
    after = float(first.detach().abs().sum())
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)
    out = ddp("train", {"pseudo_pts": torch.zeros(3, 3)}, cos_anneal_ratio=1.0, step=3)
    loss = loss_of(out, scale=float(rank + 1))              # rank-dependent gradients: 1x and 2x
    opt.zero_grad()
    loss.backward()
    g_is = float(first.grad.flatten()[0])
    g_fpn = float(next(model.feature_network.parameters()).grad.flatten()[0])
    print("RESULT " + json.dumps({"rank": rank, "before": before, "after": after, "g_is": g_is, "g_fpn": g_fpn, "calls": calls}))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_ddp_wrap_averages_the_returned_gradients_over_two_gloo_ranks(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\n" % ROOT + DDP_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    procs = [subprocess.Popen([sys.executable, str(script)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    res = [json.loads([ln for ln in o[0].splitlines() if ln.startswith("RESULT ")][-1][7:]) for o in outs]
    res.sort(key=lambda r: r["rank"])
    assert res[0]["before"] != res[1]["before"] and res[0]["after"] == res[1]["after"] == res[0]["before"]
    R = 6
    # rank r's local gradient is (r + 1) * 2 * R * 3 on the implicit surface, (r + 1) * 64 on the FPN: DDP leaves the mean
    assert res[0]["g_is"] == res[1]["g_is"] == 1.5 * 2 * R * 3
    assert res[0]["g_fpn"] == res[1]["g_fpn"] == 1.5 * 64
    assert res[0]["calls"] == ["build_fwd", "render_fwd", "render_bwd", "build_bwd"]
