"""GPU: the data-parallel training step with two processes (runner.py:102's DDP): every rank runs `training.train_step` on its
own ray batch, the gradients are averaged by `dist.all_reduce_gradients` before the optimiser step.  The box has one GPU, so
both ranks share cuda:0 and the collective runs over gloo (which moves CUDA tensors through the host) - the code path is the
one RCCL takes on N GPUs, only the transport differs.  Checks: both ranks end with identical parameters, and these equal a
single-process step on the AVERAGE of the two ranks' gradients."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from surf_amd import conf, dist as D, training
    from surf_amd.losses import Loss
    from surf_amd.surf import SuRF
    from tests.conftest import load_npz
    from tests.golden.make_golden import MODEL_CONF
    from tests.golden.make_golden_train import LOSS_CONF

    mode = sys.argv[1]                      # "dp": one of two ranks;  "ref": single process, averages both batches itself
    rank, _, world = D.init_from_env(backend="gloo") if mode == "dp" else (0, 0, 1)
    dev = torch.device("cuda:0")
    scene = load_npz("scene.npz")
    cfg = dict(MODEL_CONF)
    cfg["reg_network"] = {"d_in": [8, 16, 16, 16], "d_base": [8] * 4, "d_out": [8] * 4}
    torch.manual_seed(4)
    model = SuRF(conf.from_dict(cfg))
    with torch.no_grad():
        model.implicit_surface.deviation_network.variance.fill_(0.3)
        for net in model.reg_network.nets:
            net.out_lin.weight.mul_(4.0)
    model = model.to(dev).train()
    H, W = scene["imgs"].shape[-2:]
    R = scene["rays_o"].shape[0]

    def batch(r):                           # rank r's rays: one half of the fixture's ray lattice
        sel = torch.arange(r, R, 2)
        ipts = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in scene.items()}
        ipts["rays_o"], ipts["rays_d"] = scene["rays_o"][sel].to(dev).contiguous(), scene["rays_d"][sel].to(dev).contiguous()
        ipts["src_idx"] = 1
        g = torch.Generator().manual_seed(5 + r)
        ones = torch.ones(H, W, device=dev)
        tg = {"color": torch.rand(sel.shape[0], 3, generator=g).to(dev), "imgs": ipts["imgs"], "intrs": scene["intrs"],
              "c2ws": scene["c2ws"], "src_idx": 1, "mask_ref": ones, "mask_src": ones, "pseudo_depth_ref": ones,
              "pseudo_depth_src": ones, "depth_ref": ones, "depth_src": ones}
        return ipts, tg

    loss_fn = Loss(conf.from_dict(LOSS_CONF))
    params = model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3})
    opt = torch.optim.SGD(params, lr=1e-2)   # SGD: the update is linear in the gradient, so the reference run can average updates

    if mode == "dp":
        ipts, tg = batch(rank)
        torch.manual_seed(70)
        training.train_step(model, ipts, tg, loss_fn, opt, 1.0, 3)
    else:
        class Keep(torch.optim.SGD):         # collect the two batches' gradients, step once on their mean
            def step(self):
                pass
        keep = Keep(params, lr=1e-2)
        grads = []
        for r in range(2):
            ipts, tg = batch(r)
            torch.manual_seed(70)
            training.train_step(model, ipts, tg, loss_fn, keep, 1.0, 3)
            grads.append({n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()})
        for n, p in model.named_parameters():
            p.grad = 0.5 * (grads[0][n] + grads[1][n])
        opt.step()
    torch.cuda.synchronize()
    sd = {n: p.detach().double().cpu() for n, p in model.named_parameters() if p.requires_grad}
    out = {"sum": {n: float(v.sum()) for n, v in sd.items()}, "abs": {n: float(v.abs().sum()) for n, v in sd.items()}}
    print("RESULT " + json.dumps(out))
    D.shutdown()
""") % ROOT


def _run(args, env):
    return subprocess.Popen([sys.executable] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)


def _result(out):
    line = [ln for ln in out.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_two_rank_data_parallel_train_step(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    procs = [_run([str(script), "dp"], dict(base, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                           MASTER_PORT=str(port))) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    r0, r1 = _result(outs[0][0]), _result(outs[1][0])
    assert r0 == r1                                      # the same averaged gradient on both ranks -> identical parameters
    ref_p = _run([str(script), "ref"], base)
    ref_out = ref_p.communicate(timeout=600)
    assert ref_p.returncode == 0, ref_out[1][-2000:]
    ref = _result(ref_out[0])
    worst = 0.0
    for n, v in ref["sum"].items():
        scale = max(ref["abs"][n], 1e-9)
        worst = max(worst, abs(v - r0["sum"][n]) / scale)
    assert worst < 1e-5, worst
