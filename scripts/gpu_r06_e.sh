#!/usr/bin/env bash
set -u
O=gpurun_out/r06e; mkdir -p $O
for seed in 1 2 3 4 5; do
  for v in mfma valu; do
    if [ $v = valu ]; then export SURF_FPN_VALU=1; else unset SURF_FPN_VALU; fi
    SURF_TEST_VB_SEED=$seed python -m pytest tests/test_volume_backward.py -q -k end_to_end > $O/vb_${seed}_$v.log 2>&1; echo "seed $seed $v: $(tail -1 $O/vb_${seed}_$v.log) $(grep -o 'assert [0-9.e-]* < 0.02' $O/vb_${seed}_$v.log | head -1)"
  done
done
