#!/usr/bin/env bash
set -u
O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_hip_parity.py tests/test_autograd_runner.py tests/test_backward_fullsize.py -q -k "sdf_backward or training_backward or training_step or smooth or runner or backward" > $O/sdf.log 2>&1; echo "rc=$?" >> $O/sdf.log; tail -6 $O/sdf.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
for i in 1 2 3; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_m$i.err | tail -1 | python -c "$K" "fp32 layers-mfma"
  SURF_SDF_TRAIN_VALU=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_v$i.err | tail -1 | python -c "$K" "fp32 valu"
done
