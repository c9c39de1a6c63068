#!/usr/bin/env python3
"""Condense a scripts/profile_bench.sh output directory (gpurun_out/prof_<tag>) into profiles/<tag>_*.csv."""
import collections
import csv
import glob
import os
import sys


def newest(paths):
    """gpurun merges every call's outputs into the same local directory: take the latest run's file"""
    return max(paths, key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_digest  # noqa: E402  (ties the summary to the kernel sources it was measured on)
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def ours(name):
    """every kernel of libsurf_hip.so lives in an anonymous namespace; torch's own kernels are at::native::..."""
    return "(anonymous namespace)::" in name and "at::" not in name and "rocprim" not in name


f = newest(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")))
rows = list(csv.DictReader(open(f)))
with open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv"), "w") as w:
    w.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-seconds 0 --train-step 0 --other-configs 0   (= the default bench command without its CPU-baseline leg, without the other BASELINE configs and without the extra training-step timing, whose small launches of the same kernels would dilute the per-kernel averages; MI355X, {tag})\n")
    w.write("# surf_amd kernels verbatim; torch helper kernels (synthetic scene construction) summed in the last row\n")
    w.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev\n")
    other = 0
    for r in rows:
        if ours(r["Name"]):
            w.write(",".join('"%s"' % r[k] if k == "Name" else r[k]
                             for k in ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]) + "\n")
        else:
            other += int(r["TotalDurationNs"])
    w.write('"(all torch / runtime helper kernels)",,%d,,,,,\n' % other)

agg = collections.defaultdict(dict)
for name in ("pmc_fetch", "pmc_write", "pmc_mfma"):
    fs = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))
    if not fs:
        continue
    for row in csv.DictReader(open(newest(fs))):
        k = row["Kernel_Name"]
        if ours(k):
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[short].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_bench_pmc.csv"), "w") as w:
    w.write("# rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --mesh-grid 0 (one pass per group)\n")
    w.write("# per-launch averages; FETCH_SIZE / WRITE_SIZE in KiB as reported (raw, uncorrected)\n")
    w.write(f"# csrc_sha256: {csrc_digest()}\n")
    cols = ["FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"]
    cw = csv.writer(w)
    cw.writerow(["kernel", "launches"] + cols)
    for k, v in agg.items():
        n = max(len(x) for x in v.values())
        cw.writerow([k, n] + [str(sum(v[c]) / len(v[c])) if c in v else "" for c in cols])
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read())
print(open(os.path.join(dst, f"{tag}_bench_pmc.csv")).read())
