"""surf_amd/csrc/check_isa.py (the build-time guard of the counted `s_waitcnt vmcnt(N)` scheme) on small hand-made
assembly listings: it must accept exact and over-strict counts and refuse an under-counted wait or a scratch spill."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "surf_amd", "csrc", "check_isa.py")

REMARKS = "x.hip:1:1: remark: Function Name: _Z9my_kernelv [-Rpass]\nx.hip:1:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass]\n"


def _asm(counts):
    out = ["\t.text", "_Z9my_kernelv:"]
    for wait, n_ops in counts:
        out += ["\tbuffer_load_dwordx4 v[0:3], v4, s[0:3], 0 offen"] * n_ops
        out += ["\tv_add_f32 v0, v1, v2", f"\ts_waitcnt vmcnt({wait}) lgkmcnt(0)", "\ts_barrier"]
    out += ["\ts_endpgm", "_Z5otherv:", "\ts_waitcnt vmcnt(9)", "\ts_barrier", "\ts_endpgm"]
    return "\n".join(out) + "\n"


def _run(tmp_path, counts, scratch=0):
    a, r = tmp_path / "k.s", tmp_path / "k.remarks"
    a.write_text(_asm(counts))
    r.write_text(REMARKS % scratch)
    return subprocess.run([sys.executable, SCRIPT, str(a), str(r), "my_kernel"], capture_output=True, text=True)


def test_exact_and_stricter_counts_pass(tmp_path):
    res = _run(tmp_path, [(0, 4), (6, 6), (6, 6), (2, 9)])
    assert res.returncode == 0, res.stderr


def test_under_counted_wait_is_refused(tmp_path):
    res = _run(tmp_path, [(0, 4), (6, 6), (8, 6)])
    assert res.returncode == 1 and "vmcnt(8) but only 6" in res.stderr


def test_scratch_spill_and_too_many_loose_waits_are_refused(tmp_path):
    assert _run(tmp_path, [(0, 4), (6, 6)], scratch=16).returncode == 1
    assert _run(tmp_path, [(0, 4), (1, 6), (1, 6), (1, 6), (1, 6)]).returncode == 1
