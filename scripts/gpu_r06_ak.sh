#!/usr/bin/env bash
set -u
O=gpurun_out/r06final; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$O/bench.json').readline()); r=d['roofline']; print(d['value'], r['frac'], r['traffic'], r.get('mfma_busy'), d['training_step']['ms_per_step'])"
