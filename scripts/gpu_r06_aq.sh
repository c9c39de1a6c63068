#!/usr/bin/env bash
set -u
O=gpurun_out/r06aq; mkdir -p $O
T="import sys,json; d=json.loads(sys.stdin.read()); t=d['training_step']; print(sys.argv[1], round(t['ms_per_step'],2), t['steps_ms'], 'in-order', round(t['in_order_ms_per_step'],2))"
for g in 0 1 0 1; do
  python bench.py --force-group $g --other-configs 0 --cpu-seconds 0 --also "" --mesh-grid 64 --steps 2 --warmup 1 2>> $O/err.txt | tail -1 | python -c "$T" "default line, force-group=$g"
done
python scripts/step_phases.py 2>> $O/err.txt | tail -2
