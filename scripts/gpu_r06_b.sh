#!/usr/bin/env bash
# Round 6, second GPU run: LDS-shared weight stream of the SDF training kernels - parity tests, the full suite, the training line + its kernel stats.
set -u
O=gpurun_out/r06b; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -x -k "smooth or sdf_backward or training_backward or invalidate or conventions" > $O/sdf.log 2>&1; echo "rc=$?" >> $O/sdf.log; tail -4 $O/sdf.log
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python bench.py --cpu-seconds 5 > $O/bench.json 2> $O/bench.err; head -c 300 $O/bench.json; echo; wc -l $O/bench.json
for i in 1 2; do
  python bench.py --workload train --cpu-seconds 0 2> $O/train$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 ddp1', round(d['ms_per_step'],2))"
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_ng$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 nogroup', round(d['ms_per_step'],2))"
  python bench.py --workload train --train-precision bf16 --cpu-seconds 0 --force-group 0 2> $O/trainb$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 nogroup', round(d['ms_per_step'],2))"
done
bash scripts/profile_train.sh r06b nopmc > $O/profile_train.log 2>&1
