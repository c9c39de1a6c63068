#!/usr/bin/env bash
set -u
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -k "fpn or invalidate" > $O/fpn.log 2>&1; echo "rc=$?" >> $O/fpn.log; tail -25 $O/fpn.log
python -m pytest tests/test_hip_parity.py -q -k "sdf_backward or training_backward or training_step" > $O/sdf.log 2>&1; echo "rc=$?" >> $O/sdf.log; tail -12 $O/sdf.log
SURF_FPN_VALU=1 python -m pytest tests/test_volume_backward.py -q -k end_to_end > $O/vb_valu.log 2>&1; tail -3 $O/vb_valu.log
python -m pytest tests/test_volume_backward.py -q -k end_to_end > $O/vb_mfma.log 2>&1; tail -3 $O/vb_mfma.log
for i in 1 2; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_m$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 sdf_bwd mfma', round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
  SURF_SDF_TRAIN_VALU=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_v$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 sdf_bwd valu', round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
done
