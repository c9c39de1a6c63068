#!/usr/bin/env bash
# A/B: side streams for the backward sweep. SURF_SIDE_STREAM = 0 | comma list of users (unet,render,match,fpn) | 1 (default set) | all
set -u
O=gpurun_out/r06u; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "sparse_unet or training or volume_backward or autograd or backward or rccl or fpn or finetune" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3; do
  for m in 0 unet,render,match 1; do
    SURF_SIDE_STREAM=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "side=$m"
  done
done
python bench.py --workload train --cpu-seconds 0 --steps 10 2> $O/ddp.err | tail -1 | python -c "$K" "side=1 DDP world-1"
python bench.py --workload train --cpu-seconds 0 --steps 10 --train-precision bf16 2> $O/ddpb.err | tail -1 | python -c "$K" "side=1 DDP world-1 bf16"
python bench.py --workload train --cpu-seconds 0 --steps 10 --force-group 0 --train-precision bf16 2> $O/b.err | tail -1 | python -c "$K" "side=1 no group bf16"
python scripts/host_vs_gpu_step.py > $O/host.txt 2>&1; head -12 $O/host.txt
