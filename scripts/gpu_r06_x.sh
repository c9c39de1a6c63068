#!/usr/bin/env bash
set -u
O=gpurun_out/r06x; mkdir -p $O
python scripts/count_aten_ops.py > $O/aten.txt 2>&1; head -60 $O/aten.txt | cut -c1-150
