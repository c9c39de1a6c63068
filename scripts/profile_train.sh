#!/usr/bin/env bash
# Run on the GPU box: rocprofv3 kernel trace + stats of the training step (bench.py --workload train), then FETCH / WRITE PMC passes.
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_${1:-r03}_train"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# per-kernel statistics: in-order launches (SURF_SIDE_STREAM=0) - kernels that overlap on several streams lengthen one another, their
# durations then say nothing about the kernel; the multi-stream step is traced separately below and summarised as a timeline
export SURF_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload train --steps 5 --warmup 2 --kernel-pass 0 > "$OUT/bench_trace.log" 2>&1
if [[ "${2:-pmc}" == "pmc" ]]; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --workload train --steps 1 --warmup 1 --kernel-pass 0 > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --workload train --steps 1 --warmup 1 --kernel-pass 0 > "$OUT/bench_pmc_write.log" 2>&1
fi
unset SURF_SIDE_STREAM
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace_streams" -- python3 "$REPO/bench.py" --workload train --steps 4 --warmup 2 --kernel-pass 0 --force-group 0 --cpu-seconds 0 > "$OUT/bench_trace_streams.log" 2>&1
python3 "$REPO/scripts/trace_timeline.py" "$(find "$OUT/trace_streams" -name "*kernel_trace.csv" | head -1)" 3 > "$OUT/timeline_streams.txt" 2>&1
python3 "$REPO/scripts/trace_timeline.py" "$(find "$OUT/trace" -name "*kernel_trace.csv" | head -1)" 3 > "$OUT/timeline_inorder.txt" 2>&1
cd "$REPO"
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -45 "$f"; done
tail -1 "$OUT/bench_trace.log" | cut -c1-400
find "$OUT" -name "*.db" -delete
find "$OUT" -name "*kernel_trace.csv" -delete
du -sh "$OUT"
