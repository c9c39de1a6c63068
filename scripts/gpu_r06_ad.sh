#!/usr/bin/env bash
set -u
O=gpurun_out/r06ad; mkdir -p $O
python scripts/host_stalls.py > $O/stalls.txt 2>&1; tail -60 $O/stalls.txt
