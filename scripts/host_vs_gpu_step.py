"""Is the training step host bound?  Per step: the time the host needs to ISSUE it (until optimizer.step() returns: includes the
step's own synchronising reads - the voxel counts of the compactions, the scalar upstream gradients) against the time until the
device has finished it; then the synchronising calls of one step (torch's sync debug mode) and a cProfile of the host side with
the device far behind (kernels are asynchronous: what remains is Python + launch overhead + the forced waits).
    python scripts/host_vs_gpu_step.py [--ddp]"""
import cProfile, io, os, pstats, sys, time, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import training
from bench import training_step_setup

dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
step = lambda: training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
for _ in range(3):
    step()
torch.cuda.synchronize()
issue, total = [], []
for _ in range(8):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    issue.append((t1 - t0) * 1e3); total.append((t2 - t0) * 1e3)
print("issue ms ", " ".join(f"{x:.1f}" for x in issue))
print("total ms ", " ".join(f"{x:.1f}" for x in total))
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
import collections
sites = collections.Counter()
for x in w:
    if "synchron" in str(x.message).lower():
        sites[f"{os.path.relpath(x.filename)}:{x.lineno}"] += 1
print("synchronising calls of one step (as torch's sync debug mode reports them):")
for k, v in sites.most_common(40):
    print(f"  {v:3d}  {k}")
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:7000])
