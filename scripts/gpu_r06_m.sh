#!/usr/bin/env bash
set -u
REPO="$(pwd)"; OUT="$REPO/gpurun_out/prof_r06m"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload train --force-group 0 --cpu-seconds 0 --steps 5 --warmup 2 > "$OUT/bench.log" 2>&1
cd "$REPO"
f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
grep -E "bwd_setup|bwd_forward|bwd_reverse|bwd_scatter|sdf_smooth" $f | awk -F'"' '{split($3,a,","); printf "   %-80s calls %s ms/step %.3f avg_us %.1f min %.1f max %.1f\n", substr($2,1,80), a[2], a[3]/7e6, a[4]/1000, a[6]/1000, a[7]/1000}'
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
