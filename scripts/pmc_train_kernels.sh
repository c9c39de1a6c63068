#!/usr/bin/env bash
# SQ counters of selected kernels of the training step (bench.py --workload train --steps 1 --warmup 1), three counter passes.
#   scripts/pmc_train_kernels.sh "matching_depth_bwd|costvol_tile|costvol_bwd_kernel|ptloss_bwd_terms|blend_bwd|spconv_wgrad_thin"
PAT="${1:-matching_depth_bwd}"
REPO="$(pwd)"; OUT="$REPO/gpurun_out/pmc_train"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -- python3 "$REPO/bench.py" --workload train --steps 1 --warmup 1 --cpu-seconds 0 > "$OUT/$1.log" 2>&1; }
run p1 "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES SQ_WAVES"
run p2 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA"
run p3 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
cd "$REPO"
PAT="$PAT" python3 - <<'PY'
import csv, glob, collections, os, re
pat = re.compile(os.environ["PAT"])
rows = collections.defaultdict(lambda: collections.defaultdict(float))
for p in ('p1', 'p2', 'p3'):
    fs = glob.glob(f'gpurun_out/pmc_train/{p}/*/*counter_collection.csv')
    if not fs:
        print(p, 'no csv'); continue
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name']
        m = pat.search(k)
        if not m:
            continue
        key = re.sub(r'\(anonymous namespace\)::', '', k).split('(')[0][:60]
        rows[key][row['Counter_Name']] += float(row['Counter_Value'])          # summed over the launches of the 2 steps
        if p == 'p1' and row['Counter_Name'] == 'SQ_WAVE_CYCLES':
            rows[key]['dur_ms'] += (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e6
            rows[key]['launches'] += 1
for k, v in sorted(rows.items()):
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    print(f"{k}: {v['dur_ms']:.2f} ms over {int(v['launches'])} launches")
    for a, b in sorted(v.items()):
        if a not in ('dur_ms', 'launches'):
            print('   %-24s %.4e  (/wave_cycles %.3f)' % (a, b, b / wc))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -delete; find "$OUT" -name "*counter_collection.csv" -delete
