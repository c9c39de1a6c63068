#!/usr/bin/env bash
# matching-field / cost-volume backward timings (HIP events around the whole op, ms per training step) for the default library and
# the variants named (build_variants/*.so); matching parity tests on the default first
set -u
O=gpurun_out/${1:-md}; mkdir -p $O; shift
python -m pytest tests -m gpu -q -x -k "matching or volume_backward or training_backward or backward_fullsize" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
t() { python bench.py --workload train --cpu-seconds 0 2>> $O/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k={x['kernel']:x['ms_per_step'] for x in d['roofline_kernels']}; print('$1', 'step', round(d['ms_per_step'],2), 'matching_depth_bwd', round(k['matching_depth_bwd'],3), 'costvol_bwd', round(k['costvol_bwd'],3))"; }
t warmup; t default
for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
t default
for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
