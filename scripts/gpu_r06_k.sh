#!/usr/bin/env bash
set -u
REPO="$(pwd)"
for tag in rows0 rows1 rowsall; do
  OUT="$REPO/gpurun_out/prof_r06k_$tag"; mkdir -p "$OUT"
  case $tag in rows0) export SURF_BF16_ROWS=0;; rows1) export SURF_BF16_ROWS=1;; rowsall) export SURF_BF16_ROWS=all;; esac
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --workload train --train-precision bf16 --force-group 0 --cpu-seconds 0 --steps 5 --warmup 2 > "$OUT/bench.log" 2>&1
  cd "$REPO"
  f=$(find "$OUT/trace" -name "*kernel_stats.csv" | head -1)
  echo "== $tag total kernel ms/step: $(python3 -c "import csv,sys; r=list(csv.DictReader(open('$f'))); print(round(sum(int(x['TotalDurationNs']) for x in r)/7e6,2))")"
  grep -E "spconv_pipe_kernel<16, 8|rows_to_bf16|bn_apply_kernel|bn_bwd_apply" $f | awk -F'"' '{split($3,a,","); printf "   %-90s calls %s ms/step %.3f\n", substr($2,1,90), a[2], a[3]/7e6}'
  find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete
done
