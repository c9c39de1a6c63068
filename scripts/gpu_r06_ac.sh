#!/usr/bin/env bash
set -u
O=gpurun_out/r06ac; mkdir -p $O
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3 4; do
  for m in 0 1 -1; do
    SURF_SIDE_PRIORITY=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 20 --kernel-pass 0 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "lane priority=$m"
  done
done
python - <<'PY'
import torch, sys, os
sys.path.insert(0, os.getcwd())
from bench import training_step_setup
from surf_amd import training
dev = torch.device("cuda:0")
model, ipts, targets, loss_fn, opt = training_step_setup(dev, 576, 800, 5, 88, 512)
for i in range(40):
    training.train_step(model, ipts, targets, loss_fn, opt, 1.0, 3)
    if i in (2, 5, 10, 20, 39):
        torch.cuda.synchronize()
        print(f"step {i}: reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB, allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB, peak {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
PY
