#!/usr/bin/env bash
set -u
O=gpurun_out/r06ai; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "spconv or sparse_unet or wgrad or volume_backward or training" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
V=$PWD/build_variants/notriple.so
run() { if [[ $1 == A ]]; then SURF_HIP_LIB=$V python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 15 --kernel-pass 0 2>> $O/err.txt | tail -1 | python -c "$K" "single $2"; else python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 15 --kernel-pass 0 2>> $O/err.txt | tail -1 | python -c "$K" "12-byte $2"; fi; }
for s in 1 0; do
  export SURF_SIDE_STREAM=$s
  for o in A B B A A B B A; do run $o "streams=$s"; done
done
