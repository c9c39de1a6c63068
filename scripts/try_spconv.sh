#!/usr/bin/env bash
# Same-box A/B of the thin sparse-conv kernels: default library against the variants in build_variants/ (arguments after the
# output name).  Sparse U-Net parity tests on the default, training step three times each, kernel table of the default.
set -u
O=gpurun_out/${1:-spc}; mkdir -p $O; shift
python -m pytest tests -m gpu -q -x -k "spconv or sparse_unet or volume_build or training_backward" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -2 $O/pytest.log
t() { python bench.py --workload train --cpu-seconds 0 2>> $O/err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2))"; }
t warmup
for i in 1 2 3; do
  t default
  for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
done
bash scripts/profile_train.sh $(basename $O) nopmc > $O/profile_train.log 2>&1
