"""Which calls of the training step block the HOST?  Every public function of surf_amd.ops (and a few torch entry points) is wrapped
with a wall-clock timer; calls that take longer than 0.25 ms of host time in one step are listed with their thread (the
backward runs on autograd's thread).  Kernels are asynchronous: a long host time means a synchronising call, an allocation that
went to the driver, or a full launch queue.    python scripts/host_stalls.py"""
import os, sys, time, threading, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import ops, training
from bench import training_step_setup

dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
for _ in range(3):
    training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
torch.cuda.synchronize()
log = []
T0 = [0.0]


def wrap(mod, name, label):
    fn = getattr(mod, name)

    def timed(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = time.perf_counter() - t
            if d > 0.25e-3:
                log.append((t - T0[0], d, label, threading.current_thread().name))
    setattr(mod, name, timed)


for n, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_"):
        wrap(ops, n, "ops." + n)
for n in ("run", "join", "fork"):
    wrap(ops.side, n, "side." + n)
for n in ("zeros", "empty", "zeros_like", "cat", "full"):
    wrap(torch, n, "torch." + n)
wrap(opt, "step", "optimizer.step")
T0[0] = time.perf_counter()
training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
t_issue = time.perf_counter() - T0[0]
torch.cuda.synchronize()
print(f"step issued in {t_issue * 1e3:.1f} ms, done in {(time.perf_counter() - T0[0]) * 1e3:.1f} ms; host calls > 0.25 ms:")
for t, d, label, th in sorted(log):
    print(f"  @{t * 1e3:7.2f} ms  {d * 1e3:7.2f} ms  {label:40s} {th}")
