#!/usr/bin/env bash
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc_sq"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' > "$OUT/sq_counters.txt"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES --output-format csv -d "$OUT/p1" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --height 144 > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$OUT/p2" -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --cpu-seconds 0 --height 144 > "$OUT/p2.log" 2>&1
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
for p in ('p1','p2'):
    fs = glob.glob(f'gpurun_out/pmc_sq/{p}/*/*counter_collection.csv')
    if not fs: print(p, 'no csv'); continue
    agg = collections.defaultdict(dict)
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name']
        if 'sdf_mlp' in k or 'blend_kernel' in k:
            agg['sdf' if 'sdf_mlp' in k else 'blend'][row['Counter_Name']] = float(row['Counter_Value'])
    for k, v in agg.items(): print(p, k, {a: '%.3e' % b for a, b in v.items()})
PY
tail -3 "$OUT/p1.log" | cut -c1-300
find "$OUT" -name "*.db" -delete
