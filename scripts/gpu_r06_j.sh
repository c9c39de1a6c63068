#!/usr/bin/env bash
set -u
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -k "rows16 or sparse_unet or spconv" > $O/sp.log 2>&1; echo "rc=$?" >> $O/sp.log; tail -5 $O/sp.log
python -m pytest tests/test_volume_backward.py tests/test_autograd_runner.py tests/test_hip_configs.py -q > $O/vb.log 2>&1; echo "rc=$?" >> $O/vb.log; tail -4 $O/vb.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('spconv_dgrad')][:3])"
for i in 1 2 3; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_f$i.err | tail -1 | python -c "$K" "fp32"
  python bench.py --workload train --cpu-seconds 0 --force-group 0 --train-precision bf16 2> $O/train_b$i.err | tail -1 | python -c "$K" "bf16 rows16"
  SURF_BF16_ROWS=0 python bench.py --workload train --cpu-seconds 0 --force-group 0 --train-precision bf16 2> $O/train_v$i.err | tail -1 | python -c "$K" "bf16 no-rows16"
done
