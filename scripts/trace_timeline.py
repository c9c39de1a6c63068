#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV of the training step: with the backward sweep on several HIP streams the sum of
kernel durations no longer equals the time the chip is busy.  Prints, for the last N training steps of the trace (a step is
delimited by Adam's first multi_tensor_apply launch): wall time, the union of the kernel intervals (chip busy), the idle
remainder, the sum of durations, and the time during which >= 2 / >= 3 kernels were resident; then the largest idle gaps with
the kernels on either side, and the busy time by queue.
    python scripts/trace_timeline.py <kernel_trace.csv> [steps=5]"""
import csv, sys, collections

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
# step boundaries: the optimiser's addcmul / addcdiv launches form one cluster per step (clusters are > 20 ms apart)
clusters = []
for i, (s, e, n, q) in enumerate(rows):
    if "multi_tensor_apply" in n and "PointwiseOp" in n:      # Adam's addcmul / addcdiv (the U-Nets' counter bump is a foreach too)
        if not clusters or s - rows[clusters[-1][-1]][0] > 20_000_000:
            clusters.append([i])
        else:
            clusters[-1].append(i)
ends = [c[-1] for c in clusters]        # a step = after the previous optimiser's last launch .. this optimiser's last launch
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
print(f"{len(rows)} launches, {len(ends)} optimiser steps found")
for k in range(max(1, len(ends) - nsteps), len(ends)):
    lo, hi = ends[k - 1] + 1, ends[k]
    seg = rows[lo:hi + 1]
    t0, t1 = seg[0][0], max(e for s, e, n, q in seg)
    ev = []
    for s, e, n, q in seg:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, prev, busy, two, three = 0, t0, 0, 0, 0
    for t, d in ev:
        if depth >= 1: busy += t - prev
        if depth >= 2: two += t - prev
        if depth >= 3: three += t - prev
        depth += d; prev = t
    tot = sum(e - s for s, e, n, q in seg)
    print(f"step {k}: wall {(t1 - t0) / 1e6:7.2f} ms  busy {busy / 1e6:7.2f}  idle {(t1 - t0 - busy) / 1e6:6.2f}  sum of durations {tot / 1e6:7.2f}"
          f"  >=2 resident {two / 1e6:6.2f}  >=3 {three / 1e6:6.2f}  launches {len(seg)}")
lo, hi = ends[-2] + 1, ends[-1]
seg = rows[lo:hi + 1]
# idle gaps of the last step
gaps = []
cur_end, cur_name = seg[0][1], seg[0][2]
for s, e, n, q in seg[1:]:
    if s > cur_end:
        gaps.append((s - cur_end, cur_name, n, (cur_end - seg[0][0]) / 1e6))
    if e > cur_end:
        cur_end, cur_name = e, n
gaps.sort(reverse=True)
print("largest idle gaps of the last step (us, at ms into the step, after -> before):")
for g, a, b, at in gaps[:25]:
    print(f"  {g / 1e3:8.1f} us @ {at:6.2f} ms   {short(a)}  ->  {short(b)}")
print(f"  gaps > 20 us: {sum(1 for g in gaps if g[0] > 20000)} totalling {sum(g[0] for g in gaps if g[0] > 20000) / 1e6:.2f} ms; all gaps {sum(g[0] for g in gaps) / 1e6:.2f} ms in {len(gaps)}")
byq = collections.Counter()
for s, e, n, q in seg:
    byq[q] += e - s
print("sum of durations by queue (last step):", {q: round(v / 1e6, 2) for q, v in byq.most_common()})
