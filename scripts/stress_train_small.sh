#!/usr/bin/env bash
# Race screen of the round-5 training kernels that synchronise through atomics (masked L1's last-workgroup reduction, the cost-volume
# binning, the BN finalize, the densify gather): their parity tests N times in fresh processes; prints failures / runs.
set -u
N=${1:-12}; O=gpurun_out/stress_small; mkdir -p $O; fail=0
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q -x -k "masked_l1 or costvol or occupied or densify or sparse_unet_backward or training_backward or volume_backward" > $O/run$i.log 2>&1 || { fail=$((fail+1)); tail -5 $O/run$i.log; }
done
echo "stress: $fail failures / $N runs"; tail -1 $O/run$N.log
