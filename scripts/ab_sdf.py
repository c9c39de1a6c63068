"""Same-box A/B of split-SDF library builds: scripts/ab_sdf.py [--reps 3] [--H 576] [--stress 20] [--prec bf16x3] base NAME1 NAME2 ...
(`base` = the in-tree library, NAME = build_variants/NAME.so).  Per library, interleaved over the repetitions: time_sdf.py's
gradient- and forward-kernel times (child processes: the library is chosen at load time), then the race screen (stress_sdf.py)."""
import argparse, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--H", type=int, default=576)
ap.add_argument("--stress", type=int, default=20)
ap.add_argument("--prec", default="bf16x3")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()


def env_for(name):
    e = dict(os.environ, SURF_PREC=a.prec)
    if name != "base":
        e["SURF_HIP_LIB"] = os.path.join(ROOT, "build_variants", name + ".so")
    return e


res = {n: {"grad": [], "fwd": []} for n in a.libs}
for rep in range(a.reps):
    for n in a.libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "time_sdf.py"), str(a.H)], env=env_for(n), capture_output=True, text=True)
        for m in re.finditer(r"grad=(True|False): ([0-9.]+) ms", out.stdout):
            res[n]["grad" if m.group(1) == "True" else "fwd"].append(float(m.group(2)))
        if out.returncode != 0:
            print(n, "FAILED", out.stderr[-400:])
        for line in out.stdout.split("\n"):
            if "phases" in line and rep == a.reps - 1:
                print(n, line.strip())
for n in a.libs:
    print(f"{n:24s} grad ms {res[n]['grad']}  fwd ms {res[n]['fwd']}")
if a.stress > 0:
    for n in a.libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "stress_sdf.py"), str(a.stress), "600000", a.prec], env=env_for(n), capture_output=True, text=True)
        print(f"{n:24s} stress: {out.stdout.strip() or out.stderr[-300:]}")
