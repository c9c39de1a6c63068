#!/usr/bin/env bash
# One gpurun call while trimming the training step: full GPU tests, the training line twice per policy, the aten-op census.
set -u
O=gpurun_out/${1:-trytrain}; mkdir -p $O
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
for i in 1 2; do
  python bench.py --workload train --cpu-seconds 0 2> $O/train$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', round(d['ms_per_step'],2))"
  python bench.py --workload train --train-precision bf16 --cpu-seconds 0 2> $O/trainb$i.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16', round(d['ms_per_step'],2))"
done
python scripts/count_aten_ops.py > $O/aten_ops.txt 2>&1; grep -c . $O/aten_ops.txt
