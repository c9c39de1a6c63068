"""Static instruction mix of one kernel in a hipcc -save-temps device assembly, per basic block (scripts/isa_mix.py FILE.s KERNEL_SUBSTR).
Used to find where a VALU-issue-bound kernel spends its instructions (blend_split: per-view loops)."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ("entry", collections.Counter(), collections.Counter())
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith((";", "//", ".")) and not re.match(r"^\.LBB\S+:", s):
            continue
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            blocks.append(cur)
            cur = (m.group(1), collections.Counter(), collections.Counter())
            continue
        op = s.split()[0]
        cur[1][classify(op)] += 1
        cur[2][op] += 1
    blocks.append(cur)
    tot = collections.Counter()
    for name, c, ops in blocks:
        n = sum(c.values())
        tot.update(c)
        if n >= 40:
            print(f"{name:12s} {n:6d}  " + "  ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
            if len(sys.argv) > 3:
                print("      " + "  ".join(f"{k}={v}" for k, v in ops.most_common(int(sys.argv[3]))))
    print("total", sum(tot.values()), dict(tot))


if __name__ == "__main__":
    main()
