#!/usr/bin/env bash
# volume build (inference) timing for the default library and the variants named: bench.py's volume_build leg; densify / volume parity first
set -u
O=gpurun_out/${1:-vb}; mkdir -p $O; shift
python -m pytest tests -m gpu -q -x -k "densify or volume_build or end_to_end or training_step_forward" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
t() { python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-step 0 --mesh-grid 0 --also "" 2>> $O/err.log | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); vb=d['volume_build']; print('$1', 'build', round(vb['total_ms'],2), [ (round(s.get('densify_ms',0),2)) for s in vb.get('stages',[]) ] if isinstance(vb.get('stages'),list) else '')"; }
t warmup; t default
for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
t default
for v in "$@"; do SURF_HIP_LIB=$PWD/build_variants/$v.so t $v; done
