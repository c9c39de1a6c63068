#!/usr/bin/env bash
set -u
O=gpurun_out/r06p; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('sdf')])"
for i in 1 2 3; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_m$i.err | tail -1 | python -c "$K" "fp32 layers"
  SURF_SDF_TRAIN_VALU=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_v$i.err | tail -1 | python -c "$K" "fp32 monolithic"
done
