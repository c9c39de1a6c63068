#!/usr/bin/env bash
set -u
O=gpurun_out/r06ab; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "fpn or training or autograd or runner or sparse_unet or side_streams" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log | cut -c1-300
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3 4; do
  for m in 0 1; do
    SURF_LAYOUT_CACHE=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 --kernel-pass 0 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "layout_cache=$m"
  done
done
