#!/usr/bin/env bash
# Timeline of the multi-stream training step: rocprofv3 kernel trace -> scripts/trace_timeline.py (busy union, idle gaps, residency).
set -u
REPO="$(pwd)"; O=$REPO/gpurun_out/r06v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  SURF_SIDE_STREAM=$m rocprofv3 --kernel-trace --output-format csv -d $O/trace$m -- python3 $REPO/bench.py --workload train --steps 4 --warmup 2 --kernel-pass 0 --force-group 0 --cpu-seconds 0 > $O/bench$m.log 2>&1
  f=$(find $O/trace$m -name "*kernel_trace.csv" | head -1)
  python3 $REPO/scripts/trace_timeline.py $f 3 > $O/timeline$m.txt 2>&1
  cp $f $O/kernel_trace$m.csv; rm -rf $O/trace$m
  head -8 $O/timeline$m.txt
done
