#!/usr/bin/env bash
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc_tsdf"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/p1" -- python3 "$REPO/scripts/time_sdf.py" > "$OUT/p1.log" 2>&1
cd "$REPO"
python3 - <<'PY'
import csv, glob, collections
fs = glob.glob('gpurun_out/pmc_tsdf/p1/*/*counter_collection.csv')
rows = collections.defaultdict(dict)
for row in csv.DictReader(open(fs[0])):
    k = row['Kernel_Name']
    if 'sdf_mlp' in k:
        key = ('grad' if '<true>' in k else 'fwd', row['Dispatch_Id'])
        rows[key][row['Counter_Name']] = float(row['Counter_Value'])
        rows[key]['dur_ms'] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
for k, v in sorted(rows.items()):
    clk = v['GRBM_GUI_ACTIVE'] / 8 / (v['dur_ms'] * 1e-3) / 1e9
    busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024)
    print(k, 'dur %.1f ms clk %.2f GHz mfma_busy %.3f' % (v['dur_ms'], clk, busy), {a: '%.3e' % b for a, b in v.items() if a not in ('dur_ms',)})
PY
