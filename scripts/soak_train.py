"""Soak of the multi-stream training step: N optimiser steps on the bench scene; the loss must stay finite and fall, memory must stay flat.
    python scripts/soak_train.py [steps=300]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from surf_amd import training
from bench import training_step_setup

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
losses, t0 = [], time.perf_counter()
for i in range(N):
    out = training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
    losses.append(out["loss"])
    assert all(v == v and abs(v) < 1e30 for v in out.values()), (i, out)
    if i in (0, 9, 49, 99, 199, N - 1):
        print(f"step {i:4d}: loss {out['loss']:.4f}  colour {out['color_loss']:.4f}  photo {out['photo_loss']:.4f}  reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB  peak allocated {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = all(p.isfinite().all() for p in model.parameters())
print(f"{N} steps in {dt:.1f} s = {dt / N * 1e3:.1f} ms per step; first 10 mean {sum(losses[:10]) / 10:.4f}, last 10 mean {sum(losses[-10:]) / 10:.4f}; parameters finite: {bool(ok)}")
