#!/usr/bin/env python3
"""Condense a scripts/profile_train.sh output directory (gpurun_out/prof_<tag>_train) into profiles/<tag>_train_kernel_stats.csv
(rocprofv3 --kernel-trace --stats of `bench.py --workload train --steps 5 --warmup 2`: 7 training steps + the scene set-up) and
profiles/<tag>_train_pmc.csv (FETCH_SIZE / WRITE_SIZE per launch from separate --pmc passes of a 2-step run)."""
import collections
import csv
import glob
import os
import sys


def newest(paths):
    """gpurun merges every call's outputs into the same local directory: take the latest run's file"""
    return max(paths, key=os.path.getmtime)

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from bench import csrc_digest  # noqa: E402
src = os.path.join(root, "gpurun_out", f"prof_{tag}_train")
dst = os.path.join(root, "profiles")
STEPS = 7


def ours(name):
    return "(anonymous namespace)::" in name and "at::" not in name and "rocprim" not in name


f = newest(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")))
rows = list(csv.DictReader(open(f)))
total = sum(int(r["TotalDurationNs"]) for r in rows)
with open(os.path.join(dst, f"{tag}_train_kernel_stats.csv"), "w") as w:
    w.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload train --steps 5 --warmup 2 --kernel-pass 0 with SURF_SIDE_STREAM=0 (in-order launches: overlapped kernels lengthen one another)   (MI355X, {tag}: {STEPS} training "
            f"steps = forward + Loss + loss.backward() + Adam at 5 views 576x800, 512 rays x 128 samples, 88^3 -> 704^3 pyramid)\n")
    w.write(f"# kernel time per step: {total / STEPS / 1e6:.2f} ms (all kernels, incl. torch's fill / copy / reduce helpers, summed in the last row)\n")
    w.write(f"# csrc_sha256: {csrc_digest()}\n")
    w.write("Name,Calls,CallsPerStep,TotalDurationNs,MsPerStep,AverageNs,Percentage,MinNs,MaxNs\n")
    other = 0
    for r in rows:
        if ours(r["Name"]):
            w.write('"%s",%s,%.2f,%s,%.3f,%s,%s,%s,%s\n' % (r["Name"], r["Calls"], int(r["Calls"]) / STEPS, r["TotalDurationNs"],
                                                          int(r["TotalDurationNs"]) / STEPS / 1e6, r["AverageNs"], r["Percentage"],
                                                          r["MinNs"], r["MaxNs"]))
        else:
            other += int(r["TotalDurationNs"])
    w.write('"(all torch / runtime helper kernels)",,,%d,%.3f,,,,\n' % (other, other / STEPS / 1e6))

agg = collections.defaultdict(dict)
for name in ("pmc_fetch", "pmc_write"):
    fs = glob.glob(os.path.join(src, name, "*", "*counter_collection.csv"))
    if not fs:
        continue
    for row in csv.DictReader(open(newest(fs))):
        k = row["Kernel_Name"]
        if ours(k):
            short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            agg[short].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
if agg:
    with open(os.path.join(dst, f"{tag}_train_pmc.csv"), "w") as w:
        w.write("# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python3 bench.py --workload train --steps 1 --warmup 1 (one pass per counter)\n")
        w.write("# per-launch averages in KiB as reported (raw: double FETCH_SIZE for 16-byte-per-lane streaming reads, MI355X_MICROARCH.md)\n")
        w.write(f"# csrc_sha256: {csrc_digest()}\n")
        cw = csv.writer(w)
        cw.writerow(["kernel", "launches", "FETCH_SIZE", "WRITE_SIZE"])
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0])) - sum(kv[1].get("WRITE_SIZE", [0]))):
            n = max(len(x) for x in v.values())
            cw.writerow([k, n] + [f"{sum(v[c]) / len(v[c]):.1f}" if c in v else "" for c in ("FETCH_SIZE", "WRITE_SIZE")])
print(open(os.path.join(dst, f"{tag}_train_kernel_stats.csv")).read()[:3000])
if agg:
    print(open(os.path.join(dst, f"{tag}_train_pmc.csv")).read()[:2500])
