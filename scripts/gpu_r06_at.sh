#!/usr/bin/env bash
set -u
O=gpurun_out/r06at; mkdir -p $O
T="import sys,json; d=json.loads(sys.stdin.read()); t=d['training_step']; print(sys.argv[1], round(t['ms_per_step'],2), 'ddp', round(t['ddp_ms_per_step'] or 0,2), 'in-order', round(t['in_order_ms_per_step'],2))"
for l in 1 0 0 1 1 0; do
  SURF_SMOOTH_LANE=$l python bench.py --other-configs 0 --cpu-seconds 0 --also "" --mesh-grid 64 --steps 2 --warmup 1 2>> $O/err.txt | tail -1 | python -c "$T" "group, smooth lane=$l"
done
