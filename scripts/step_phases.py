"""Where a training step's wall time goes, phase by phase, on the host clock and on the device clock (events on the main stream):
forward (model), loss, loss.backward(), optimizer.step().  A phase whose device time exceeds the sum of its kernels is waiting for
the host.    python scripts/step_phases.py [fused] [ddp [nobuf] [bucketview]]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import training_step_setup

dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
if "fused" in sys.argv:
    opt = torch.optim.Adam(model.get_optim_params({"mlp_lr": 5e-4, "feat_lr": 1e-3}), fused=True)
stepper = model
if "ddp" in sys.argv:
    from surf_amd import dist as D
    D.init_from_env("nccl", dev, force=True, timeout_s=300)
    kw = {"broadcast_buffers": False} if "nobuf" in sys.argv else {}
    if "bucketview" in sys.argv:
        kw["gradient_as_bucket_view"] = True
    stepper = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], **kw)             # runner.py:102
inputs = {**targets, **ipts}
names = ["forward", "loss", "backward", "optimizer"]


def step(rec=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
    th = [time.perf_counter()]
    ev[0].record()
    out = stepper("train", inputs, cos_anneal_ratio=1.0, step=3.0)
    ev[1].record(); th.append(time.perf_counter())
    loss = loss_fn(out, inputs, 3.0)["loss"]
    ev[2].record(); th.append(time.perf_counter())
    opt.zero_grad(set_to_none=True)
    loss.backward()
    ev[3].record(); th.append(time.perf_counter())
    opt.step()
    ev[4].record(); th.append(time.perf_counter())
    torch.cuda.synchronize()
    th.append(time.perf_counter())
    if rec is not None:
        rec.append(([ev[i].elapsed_time(ev[i + 1]) for i in range(4)], [(th[i + 1] - th[i]) * 1e3 for i in range(5)]))


for _ in range(3):
    step()
rec = []
for _ in range(8):
    step(rec)
med = lambda xs: sorted(xs)[len(xs) // 2]
print("phase        device ms   host ms (issue)")
for i, n in enumerate(names):
    print(f"{n:12s} {med([r[0][i] for r in rec]):8.2f}   {med([r[1][i] for r in rec]):8.2f}")
print(f"final sync                {med([r[1][4] for r in rec]):8.2f}")
print(f"step total   {med([sum(r[0]) for r in rec]):8.2f}   {med([sum(r[1]) for r in rec]):8.2f}")
