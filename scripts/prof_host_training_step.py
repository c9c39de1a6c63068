"""cProfile of the host side of one full training step (after warm-up)."""
import cProfile, pstats, runpy, sys, os, io
sys.argv = [sys.argv[0]]
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_full_training_step.py"))
import torch
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
ns["training"].train_step(ns["model"], ns["ipts"], ns["targets"], ns["loss_fn"], ns["opt"], 1.0, 3)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
