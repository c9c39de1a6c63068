#!/usr/bin/env python3
"""Build-time check of the kernels that retire LDS-DMA with a counted `s_waitcnt vmcnt(N)` (sdf_mlp_split.hip, see
stage_barrier there): the count is a compile-time function of the chunk and assumes that the compiler emits exactly
the vector-memory operations the source issues between two workgroup barriers.  This script reads the device assembly
(hipcc --cuda-device-only -S) and the resource-usage remarks of the same compile and refuses the build when

  * a checked kernel spills to scratch memory (spill loads / stores are uncounted, FLAT-class operations), or
  * any `s_waitcnt vmcnt(N)` that directly precedes an `s_barrier` has FEWER than N vector-memory instructions in the
    W barrier intervals before it (W = the `; surf_ring_window W` comment stage_barrier emits with the wait: ring length
    - 2; 1 when absent): the wait would then leave operations of an older chunk - possibly the DMA of the chunk about to
    be read - in flight (more than N is stricter than needed and allowed at the few seams where the source says so - the
    prologue, the first chunk of a round (which follows the gather) and the forward -> backward seam, each of which
    falls into W windows: at most `--max-loose` (default 3 W) barriers per kernel).

usage: check_isa.py <file.s> <remarks.txt> <kernel name substring> [--max-loose K]
"""
import re
import sys


def main():
    asm_path, remarks_path, needle = sys.argv[1:4]
    max_loose_arg = int(sys.argv[sys.argv.index("--max-loose") + 1]) if "--max-loose" in sys.argv else None
    lines = open(asm_path).read().split("\n")
    heads = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    heads.append(len(lines))
    vm = re.compile(r"^\s+(buffer_|global_|flat_|scratch_)")
    ok = True
    n_kernels = 0
    for k in range(len(heads) - 1):
        name = lines[heads[k]].split(":")[0]
        if needle not in name:
            continue
        n_kernels += 1
        cnt, last_wait, pairs, window, hist, max_w = 0, None, [], 1, [], 1
        for i in range(heads[k], heads[k + 1]):
            l = lines[i]
            if vm.match(l):
                cnt += 1
            m = re.search(r"surf_ring_window (\d+)", l)
            if m:
                window = int(m.group(1))
                max_w = max(max_w, window)
            m = re.search(r"s_waitcnt vmcnt\((\d+)\)", l)
            if m:
                last_wait = (int(m.group(1)), i)
            if re.match(r"^\s+s_barrier", l):
                hist.append(cnt)
                counted_wait = last_wait is not None and last_wait[1] == i - 1
                # straight-line position only: the round loop's back edge (first chunks of a round) sees the prologue's
                # intervals here, which hold no fewer operations than the tail of a round - those barriers are "loose"
                pairs.append((last_wait[0] if counted_wait else None, sum(hist[-window:]), i + 1))
                cnt, window = 0, 1
        max_loose = max_loose_arg if max_loose_arg is not None else 3 * max_w
        counted = [p for p in pairs if p[0] is not None]
        under = [p for p in counted if p[1] < p[0]]
        loose = [p for p in counted if p[1] > p[0]]
        print(f"check_isa: {name}: {len(counted)} counted barriers, {len(loose)} stricter than needed, {len(under)} under-counted")
        if not counted:
            print(f"check_isa: {name}: no `s_waitcnt vmcnt(N); s_barrier` pair found (was the kernel restructured?)", file=sys.stderr)
            ok = False
        for n, c, ln in under:
            print(f"check_isa: {name}: line {ln}: vmcnt({n}) but only {c} vector-memory instructions in the barrier "
                  f"intervals it spans", file=sys.stderr)
            ok = False
        if len(loose) > max_loose:
            print(f"check_isa: {name}: {len(loose)} barriers wait for more than they need (allowed: {max_loose}): "
                  f"{[(n, c, ln) for n, c, ln in loose]}", file=sys.stderr)
            ok = False
    if n_kernels == 0:
        print(f"check_isa: no kernel matching {needle!r} in {asm_path}", file=sys.stderr)
        ok = False
    # scratch spills (from -Rpass-analysis=kernel-resource-usage)
    rem = open(remarks_path).read().split("\n")
    cur = None
    seen = 0
    for l in rem:
        m = re.search(r"Function Name: (\S+)", l)
        if m:
            cur = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", l)
        if m and cur and needle in cur:
            seen += 1
            if int(m.group(1)) != 0:
                print(f"check_isa: {cur} spills {m.group(1)} bytes/lane to scratch memory", file=sys.stderr)
                ok = False
    if seen == 0:
        print("check_isa: resource usage of the checked kernels could not be read", file=sys.stderr)
        ok = False
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
