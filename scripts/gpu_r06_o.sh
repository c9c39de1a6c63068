#!/usr/bin/env bash
REPO="$(pwd)"
python -m pytest tests/test_hip_parity.py tests/test_autograd_runner.py tests/test_backward_fullsize.py -q -k "smooth or sdf_backward or training_backward or runner or backward" 2>&1 | tail -3
python scripts/time_sdf_train.py 2>&1 | tail -1
SURF_SDF_TRAIN_VALU=1 python scripts/time_sdf_train.py 2>&1 | tail -1
OUT="$REPO/gpurun_out/prof_r06o"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/scripts/time_sdf_train.py" > "$OUT/log" 2>&1
cd "$REPO"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
grep -E "bwd_|sm_" $f | awk -F'"' '{split($3,a,","); printf "   %-40s avg_us %.1f\n", substr($2,26,40), a[4]/1000}' | sort
rm -rf "$OUT"
