#!/usr/bin/env bash
set -u
O=gpurun_out/r06ar; mkdir -p $O
T="import sys,json; d=json.loads(sys.stdin.read()); t=d['training_step']; print(sys.argv[1], round(t['ms_per_step'],2), 'ddp', round(t['ddp_ms_per_step'],2), 'in-order', round(t['in_order_ms_per_step'],2), '| rays/s', round(d['value']), 'scene', round(d['scene']['scene_ms'],1))"
for q in 4 8 4 8 16; do
  GPU_MAX_HW_QUEUES=$q python bench.py --other-configs 0 --cpu-seconds 0 --also "" --mesh-grid 64 --steps 2 --warmup 1 2>> $O/err.txt | tail -1 | python -c "$T" "GPU_MAX_HW_QUEUES=$q"
done
