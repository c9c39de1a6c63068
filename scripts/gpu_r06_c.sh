#!/usr/bin/env bash
# Round 6, third GPU run: the FPN on the matrix cores (parity vs the VALU kernels, goldens), the fixed tests, training line A/B.
set -u
O=gpurun_out/r06c; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -x -k "fpn or invalidate" > $O/fpn.log 2>&1; echo "rc=$?" >> $O/fpn.log; tail -12 $O/fpn.log
python -m pytest tests/test_volume_backward.py tests/test_autograd_runner.py -q -x > $O/vb.log 2>&1; echo "rc=$?" >> $O/vb.log; tail -4 $O/vb.log
python -m pytest tests/test_hip_configs.py tests/test_end_to_end_dtu.py -q -x > $O/cfg.log 2>&1; echo "rc=$?" >> $O/cfg.log; tail -4 $O/cfg.log
for i in 1 2 3; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_m$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 mfma-fpn', round(d['ms_per_step'],2))"
  SURF_FPN_VALU=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_v$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 valu-fpn', round(d['ms_per_step'],2))"
  python bench.py --workload train --cpu-seconds 0 --force-group 0 --train-precision bf16 2> $O/train_b$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 mfma-fpn', round(d['ms_per_step'],2))"
  SURF_FPN_WGRAD_THIN_MFMA=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_t$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 mfma-fpn + thin wgrad', round(d['ms_per_step'],2))"
done
bash scripts/profile_train.sh r06c nopmc > $O/profile_train.log 2>&1
