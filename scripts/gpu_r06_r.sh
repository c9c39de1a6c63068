#!/usr/bin/env bash
# A/B: the sparse U-Net's weight gradients on a side stream (SURF_SIDE_STREAM=1, default) vs in-order launches (=0).
set -u
O=gpurun_out/r06r; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "sparse_unet or training_backward or volume_backward or autograd or spconv_backward or training_step" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2), d.get('steps_ms'))"
for i in 1 2 3; do
  SURF_SIDE_STREAM=0 python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 2> $O/t0_$i.err | tail -1 | python -c "$K" "in-order   "
  SURF_SIDE_STREAM=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 2> $O/t1_$i.err | tail -1 | python -c "$K" "side stream"
done
