#!/usr/bin/env bash
set -u
O=gpurun_out/r06ap; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "volume_backward or training or side_streams or autograd or costvol or runner" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for o in A B B A A B B A; do
  if [[ $o == A ]]; then m=unet,render,match; else m=1; fi
  SURF_SIDE_STREAM=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 15 --kernel-pass 0 2>> $O/err.txt | tail -1 | python -c "$K" "train side=$m"
done
