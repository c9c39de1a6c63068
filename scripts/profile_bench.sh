#!/usr/bin/env bash
# Run on the GPU box: rocprofv3 kernel trace + stats of bench.py, then two PMC passes (FETCH_SIZE, WRITE_SIZE).
set -u
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_${1:-r01}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --cpu-seconds 0 --train-step 0 --other-configs 0 > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --mesh-grid 0 --train-step 0 --other-configs 0 --force-group 0 > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --mesh-grid 0 --train-step 0 --other-configs 0 --force-group 0 > "$OUT/bench_pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --mesh-grid 0 --train-step 0 --other-configs 0 --force-group 0 > "$OUT/bench_pmc_mfma.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -30
for f in $(find "$OUT/trace" -name "*kernel_stats.csv"); do echo "== $f"; head -12 "$f"; done
tail -2 "$OUT/bench_trace.log"
# keep the merge small: drop the raw per-dispatch traces except pmc ones
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
