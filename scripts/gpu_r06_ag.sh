#!/usr/bin/env bash
set -u
O=gpurun_out/r06ag; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "spconv or sparse_unet or pipeline or end_to_end or volume or training" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
P="import sys,json; d=json.loads(sys.stdin.read()); vb=d['volume_build']; print(sys.argv[1], 'build', round(vb['total_ms'],2), 'unet per stage', [round(s['sparse_unet_ms'],3) for s in vb['stages']])"
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
V=$PWD/build_variants/notriple.so
for i in 1 2 3; do
  SURF_HIP_LIB=$V python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-step 0 --other-configs 0 --also "" --mesh-grid 64 2> $O/b0_$i.err | tail -1 | python -c "$P" "single loads"
  python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-step 0 --other-configs 0 --also "" --mesh-grid 64 2> $O/b1_$i.err | tail -1 | python -c "$P" "12-byte    "
done
for i in 1 2 3; do
  SURF_HIP_LIB=$V python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 --kernel-pass 0 2> $O/t0_$i.err | tail -1 | python -c "$K" "train single loads"
  python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 --kernel-pass 0 2> $O/t1_$i.err | tail -1 | python -c "$K" "train 12-byte    "
done
