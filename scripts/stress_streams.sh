#!/usr/bin/env bash
# Race screen of the multi-stream backward sweep (ops.SideStream): the tests that run a whole training step or its two backward halves
# against the reference's gradients, N times in fresh processes; prints failures / runs.
set -u
N=${1:-8}; O=gpurun_out/stress_streams; mkdir -p $O; fail=0
for i in $(seq 1 $N); do
  python -m pytest tests -m gpu -q -x -k "side_streams or training_backward or volume_backward or autograd_runner or training_step or rccl_training or render_backward" > $O/run$i.log 2>&1 || { fail=$((fail+1)); tail -5 $O/run$i.log; }
done
echo "stress: $fail failures / $N runs"; tail -1 $O/run$N.log
