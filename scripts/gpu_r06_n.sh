#!/usr/bin/env bash
python -m pytest tests/test_hip_parity.py tests/test_autograd_runner.py -q -k "sdf_backward or training_backward or runner" 2>&1 | tail -2
python scripts/time_sdf_train.py 2>&1 | tail -1
SURF_SDF_TRAIN_VALU=1 python scripts/time_sdf_train.py 2>&1 | tail -1
for v in 1 2 4 15; do SURF_HIP_LIB=$PWD/build_variants/tm$v.so python scripts/time_sdf_train.py 2>&1 | tail -1; done
bash scripts/gpu_r06_m.sh
