#!/usr/bin/env bash
set -u
O=gpurun_out/r06f; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -k "sdf_backward or training_backward or training_step or smooth" > $O/sdf.log 2>&1; echo "rc=$?" >> $O/sdf.log; tail -5 $O/sdf.log
for i in 1 2; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_m$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 sdf_bwd mfma', round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
  SURF_SDF_TRAIN_VALU=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_v$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 sdf_bwd valu', round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
done
