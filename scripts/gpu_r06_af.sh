#!/usr/bin/env bash
set -u
O=gpurun_out/r06af; mkdir -p $O
python -m pytest tests -m gpu -q -x -k "matching or pipeline or golden or end_to_end or volume or surf_forward or training or tnt or config" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
P="import sys,json; d=json.loads(sys.stdin.read()); vb=d['volume_build']; print(sys.argv[1], 'build', round(vb['total_ms'],2), 'matching per stage', [round(s['matching_field_ms'],3) for s in vb['stages']])"
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3; do
  for m in 0 1; do
    SURF_MD_FORM=$m python bench.py --steps 2 --warmup 1 --cpu-seconds 0 --train-step 0 --other-configs 0 --also "" --mesh-grid 64 2> $O/b_${m}_$i.err | tail -1 | python -c "$P" "form=$m"
  done
done
for i in 1 2 3; do
  for m in 0 1; do
    SURF_MD_FORM=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 --kernel-pass 0 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "train form=$m"
  done
done
