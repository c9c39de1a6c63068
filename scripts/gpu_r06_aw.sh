#!/usr/bin/env bash
set -u
O=gpurun_out/r06aw; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "side_streams or training or autograd or runner or loss or photometric or volume_backward or rccl" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
T="import sys,json; d=json.loads(sys.stdin.read()); t=d['training_step']; print(sys.argv[1], round(t['ms_per_step'],2), 'ddp', round(t['ddp_ms_per_step'] or 0,2), 'in-order', round(t['in_order_ms_per_step'],2))"
for m in 0 1 1 0 0 1; do
  SURF_LANE_NODES=$m python bench.py --other-configs 0 --cpu-seconds 0 --also "" --mesh-grid 64 --steps 2 --warmup 1 2>> $O/err.txt | tail -1 | python -c "$T" "lane nodes=$m"
done
python scripts/step_phases.py 2>> $O/err.txt | tail -7
