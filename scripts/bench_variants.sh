#!/usr/bin/env bash
# run bench.py once per library variant in build_variants/ (A/B of kernel builds on one device)
for lib in build_variants/*.so; do
  echo "== $lib"
  SURF_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('rays/s %.0f  ms/step %.1f  kernels %s  frac %.3f' % (r['value'], r['ms_per_step'], {k: round(v, 1) for k, v in r['kernel_ms'].items()}, r['roofline']['frac']))"
done
