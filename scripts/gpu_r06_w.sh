#!/usr/bin/env bash
# Full GPU suite on the multi-stream backward (depth tap, device-side upstream scalars, ABI 39), A/B against in-order launches, timeline.
set -u
REPO="$(pwd)"; O=$REPO/gpurun_out/r06w; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
for i in 1 2 3; do
  for m in 0 1; do
    SURF_SIDE_STREAM=$m python bench.py --workload train --cpu-seconds 0 --force-group 0 --steps 10 2> $O/t_${m}_$i.err | tail -1 | python -c "$K" "side=$m"
  done
done
python bench.py --workload train --cpu-seconds 0 --steps 10 2> $O/ddp.err | tail -1 | python -c "$K" "side=1 DDP world-1"
python bench.py --workload train --cpu-seconds 0 --steps 10 --force-group 0 --train-precision bf16 2> $O/b.err | tail -1 | python -c "$K" "side=1 no group bf16"
python scripts/host_vs_gpu_step.py > $O/host.txt 2>&1; head -22 $O/host.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace1 -- python3 $REPO/bench.py --workload train --steps 4 --warmup 2 --kernel-pass 0 --force-group 0 --cpu-seconds 0 > $O/bench1.log 2>&1
f=$(find $O/trace1 -name "*kernel_trace.csv" | head -1)
python3 $REPO/scripts/trace_timeline.py $f 3 > $O/timeline1.txt 2>&1
cp $f $O/kernel_trace1.csv; rm -rf $O/trace1
head -6 $O/timeline1.txt
