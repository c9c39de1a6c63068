#!/usr/bin/env bash
# SQ counters of the kernels whose name contains $1 while running "python3 $2 ..." (three rocprofv3 --pmc passes).
# usage on the GPU box: bash scripts/pmc_kernel.sh blend_split scripts/time_blend.py 288   (env passes through)
REPO="$(pwd)"; PAT="$1"; shift
OUT="$REPO/gpurun_out/pmc_${PAT}"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/p1" -- python3 "$REPO/$1" "${@:2}" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- python3 "$REPO/$1" "${@:2}" > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/p3" -- python3 "$REPO/$1" "${@:2}" > "$OUT/p3.log" 2>&1
cd "$REPO"
PAT="$PAT" python3 - <<'PY'
import csv, glob, collections, os
pat = os.environ["PAT"]
rows = collections.defaultdict(dict)
for p in ('p1', 'p2', 'p3'):
    fs = glob.glob(f'gpurun_out/pmc_{pat}/{p}/*/*counter_collection.csv')
    if not fs:
        print(p, 'no csv'); continue
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name']
        if pat in k:
            key = k.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            rows[key][row['Counter_Name']] = float(row['Counter_Value'])       # last dispatch wins
            rows[key]['dur_ms'] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
for k, v in sorted(rows.items()):
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, 'dur %.2f ms' % v['dur_ms'])
    if 'GRBM_GUI_ACTIVE' in v:
        print('   clk %.2f GHz  mfma_busy %.3f' % (v['GRBM_GUI_ACTIVE'] / 8 / (v['dur_ms'] * 1e-3) / 1e9, v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)))
    for a, b in sorted(v.items()):
        if a != 'dur_ms':
            print('   %-28s %.4e  (/wave_cycles %.3f)' % (a, b, b / wc))
PY
tail -4 "$OUT/p1.log"
find "$OUT" -name "*.db" -delete
