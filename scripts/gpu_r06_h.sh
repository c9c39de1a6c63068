#!/usr/bin/env bash
set -u
O=gpurun_out/r06h; mkdir -p $O
python -m pytest tests/test_hip_parity.py -q -k "spconv or sparse_unet" > $O/sp.log 2>&1; echo "rc=$?" >> $O/sp.log; tail -5 $O/sp.log
SURF_THIN_MFMA=all python -m pytest tests/test_hip_parity.py tests/test_volume_backward.py -q -k "spconv or sparse_unet or end_to_end or volume" > $O/sp_all.log 2>&1; echo "rc=$?" >> $O/sp_all.log; tail -5 $O/sp_all.log
K="import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['ms_per_step'],2), [ (e['kernel'], round(e['ms_per_step'],2)) for e in d['roofline_kernels'] if e['kernel'].startswith('spconv_dgrad')][:4])"
for i in 1 2 3; do
  python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_f$i.err | tail -1 | python -c "$K" "fp32 thin-valu"
  SURF_THIN_MFMA=all python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_fa$i.err | tail -1 | python -c "$K" "fp32 thin-mfma"
  python bench.py --workload train --cpu-seconds 0 --force-group 0 --train-precision bf16 2> $O/train_b$i.err | tail -1 | python -c "$K" "bf16 thin-mfma"
  SURF_THIN_MFMA=none python bench.py --workload train --cpu-seconds 0 --force-group 0 --train-precision bf16 2> $O/train_v$i.err | tail -1 | python -c "$K" "bf16 thin-valu"
  SURF_SPCONV_THIN_OLD=1 python bench.py --workload train --cpu-seconds 0 --force-group 0 2> $O/train_o$i.err | tail -1 | python -c "$K" "fp32 (16,16 on the old mfma kernel)"
done
python bench.py --cpu-seconds 0 --mesh-grid 0 --train-step 0 --other-configs 0 --also "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('volume build valu', d['volume_build']['total_ms'])"
SURF_THIN_MFMA=all python bench.py --cpu-seconds 0 --mesh-grid 0 --train-step 0 --other-configs 0 --also "" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('volume build thin-mfma', d['volume_build']['total_ms'])"
