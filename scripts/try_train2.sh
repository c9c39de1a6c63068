#!/usr/bin/env bash
set -u
O=gpurun_out/${1:-trytrain}; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
bash scripts/profile_train.sh ${1:-trytrain} nopmc > $O/profile_train.log 2>&1
for i in 1 2; do
  python bench.py --workload train --cpu-seconds 0 2> $O/train$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', round(d['ms_per_step'],2))"
  python bench.py --workload train --train-precision bf16 --cpu-seconds 0 2> $O/trainb$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', round(d['ms_per_step'],2))"
done
