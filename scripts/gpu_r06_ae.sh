#!/usr/bin/env bash
set -u
O=gpurun_out/r06ae; mkdir -p $O
for m in "" "ddp" "ddp nobuf"; do
  echo "== step_phases $m"; python scripts/step_phases.py $m 2> $O/err.txt | tail -7
done
