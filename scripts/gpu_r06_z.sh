#!/usr/bin/env bash
set -u
O=gpurun_out/r06z; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "side_streams or photometric or autograd or training or runner or loss" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3; do
  for m in unet,render,match 1; do
    SURF_SIDE_STREAM=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 --kernel-pass 0 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "side=$m"
  done
done
