#!/usr/bin/env bash
# costvol_bwd (HIP events around the whole op, ms per training step) for the default library and the variants named
set -u
O=gpurun_out/${1:-cvx}; mkdir -p $O; shift
t() { python bench.py --workload train --cpu-seconds 0 2>> $O/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=[x for x in d['roofline_kernels'] if x['kernel']=='costvol_bwd'][0]; print('$1', 'step', round(d['ms_per_step'],2), 'costvol_bwd', round(k['ms_per_step'],3))"; }
t warmup
t default
for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
t default
