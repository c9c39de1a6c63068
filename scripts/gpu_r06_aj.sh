#!/usr/bin/env bash
set -u
O=gpurun_out/r06aj; mkdir -p $O
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2))"
run() { SURF_BF16_ROWS=$1 python bench.py --workload train --train-precision $2 --cpu-seconds 0 --force-group 0 --steps 15 --kernel-pass 0 2>> $O/err.txt | tail -1 | python -c "$K" "$2 rows16=$1"; }
for o in 0 1 1 0 0 1 1 0; do run $o bf16; done
for o in 1 1 1; do run $o fp32; done
