"""How long does the main stream wait at each join of the multi-stream backward sweep?  ops.side.join is wrapped with a HIP event pair
on the current stream (the waits are stream-side: the pair brackets exactly the time the stream spends blocked on the lanes).
    python scripts/join_stalls.py"""
import collections, os, sys, traceback
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import ops, training
from bench import training_step_setup

dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
for _ in range(3):
    training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
torch.cuda.synchronize()
rec = []
orig = ops.side.join


def join(lanes=None):
    site = [f for f in traceback.extract_stack(limit=4) if "surf_amd" in f.filename][-1]
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    orig(lanes)
    b.record()
    rec.append((f"{os.path.basename(site.filename)}:{site.lineno} lanes={lanes}", a, b))


ops.side.join = join
N = 6
for _ in range(N):
    training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
torch.cuda.synchronize()
tot = collections.defaultdict(float)
for k, a, b in rec:
    tot[k] += a.elapsed_time(b) / N
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{v:7.3f} ms per step blocked at  {k}")
