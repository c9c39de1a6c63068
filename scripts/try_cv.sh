#!/usr/bin/env bash
# Same-box A/B of the binned costvol backward (default) against the direct scatter (build_variants/cv_scatter.so).
set -u
O=gpurun_out/${1:-cv}; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "costvol or volume_backward or training_backward or backward_fullsize or autograd" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
t() { python bench.py --workload train --cpu-seconds 0 2>> $O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2))"; }
t warmup
for i in 1 2 3; do
  t default
  SURF_HIP_LIB=$PWD/build_variants/cv_scatter.so t cv_scatter
done
bash scripts/profile_train.sh $(basename $O) nopmc > $O/profile_train.log 2>&1
