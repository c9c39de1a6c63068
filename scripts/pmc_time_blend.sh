#!/usr/bin/env bash
# SQ counters of the blending kernel under scripts/time_blend.py (SURF_BLEND selects the kernel: bf16x3 | f16x2 | f32);
# three rocprofv3 counter passes, same counter set as scripts/pmc_time_sdf.sh.  usage: SURF_BLEND=bf16x3 bash scripts/pmc_time_blend.sh [H]
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc_tblend"; rm -rf "$OUT"; mkdir -p "$OUT"
export SURF_BLEND="${SURF_BLEND:-bf16x3}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/p1" -- python3 "$REPO/scripts/time_blend.py" "$@" > "$OUT/p1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d "$OUT/p2" -- python3 "$REPO/scripts/time_blend.py" "$@" > "$OUT/p2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM --output-format csv -d "$OUT/p3" -- python3 "$REPO/scripts/time_blend.py" "$@" > "$OUT/p3.log" 2>&1
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(dict)
for p in ('p1', 'p2', 'p3'):
    fs = glob.glob(f'gpurun_out/pmc_tblend/{p}/*/*counter_collection.csv')
    if not fs:
        print(p, 'no csv'); continue
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name']
        if 'blend' in k and 'pack' not in k:
            key = k.split('(')[0][:60]
            rows[key][row['Counter_Name']] = float(row['Counter_Value'])   # last dispatch wins (3 timed runs)
            rows[key]['dur_ms'] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
for k, v in sorted(rows.items()):
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(k, 'dur %.1f ms' % v['dur_ms'])
    if 'GRBM_GUI_ACTIVE' in v:
        print('   clk %.2f GHz  mfma_busy %.3f' % (v['GRBM_GUI_ACTIVE'] / 8 / (v['dur_ms'] * 1e-3) / 1e9, v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)))
    for a, b in sorted(v.items()):
        if a != 'dur_ms':
            print('   %-28s %.4e  (/wave_cycles %.3f)' % (a, b, b / wc))
PY
find "$OUT" -name "*.db" -delete
