#!/usr/bin/env python3
"""After scripts/final_r06.sh (gpurun merged its outputs into gpurun_out/): copy the judged artefacts into profiles/r06_* and
print the numbers DESIGN.md / README.md quote (python scripts/collect_r06.py [--fill] replaces the @PLACEHOLDERS@ of DESIGN.md)."""
import csv, glob, json, os, re, shutil, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
sys.path.insert(0, ROOT)
from bench import csrc_digest
O = "gpurun_out/r06final"
subprocess.run([sys.executable, "scripts/summarize_profile.py", "r06"], check=True)
subprocess.run([sys.executable, "scripts/summarize_train_profile.py", "r06"], check=True)
line = lambda p: json.loads(open(p).readline())
b = line(f"{O}/bench.json")
shutil.copy(f"{O}/bench.json", "profiles/r06_bench.json")
tr = [line(f"{O}/train{i}.json") for i in (1, 2, 3)]
tb = [line(f"{O}/train_bf16_{i}.json") for i in (1, 2, 3)]
med = lambda xs: sorted(xs, key=lambda t: t["ms_per_step"])[len(xs) // 2]
json.dump(med(tr), open("profiles/r06_bench_train.json", "w"))
json.dump(med(tb), open("profiles/r06_bench_train_bf16.json", "w"))
shutil.copy(f"{O}/scene_parts.log", "profiles/r06_scene_parts.txt")
tl = "gpurun_out/prof_r06_train"
if os.path.exists(f"{tl}/timeline_streams.txt"):
    with open("profiles/r06_train_timeline.txt", "w") as f:
        f.write("# scripts/trace_timeline.py on rocprofv3 --kernel-trace of `bench.py --workload train` (scripts/profile_train.sh; the profiler slows\n"
                "# the host: walls are larger than unprofiled).  First the default multi-stream backward sweep, then SURF_SIDE_STREAM=0.\n")
        f.write("## backward sweep on several HIP streams (default)\n" + open(f"{tl}/timeline_streams.txt").read())
        f.write("## in-order launches (SURF_SIDE_STREAM=0)\n" + open(f"{tl}/timeline_inorder.txt").read())
for name in ("phases.txt", "phases_inorder.txt"):
    if os.path.exists(f"{O}/{name}"):
        shutil.copy(f"{O}/{name}", f"profiles/r06_train_{name}")
shutil.copy(f"{O}/sdf_sq_bf16x3.txt", "profiles/r06_sdf_sq_bf16x3.txt")
with open("profiles/r06_gpu_tests.txt", "w") as f:
    f.write("# python -m pytest tests -m gpu -q ; smoke()   (scripts/final_r06.sh on the GPU box)\n")
    f.write("".join(open(f"{O}/pytest.log").readlines()[-4:]))
    f.write(open(f"{O}/smoke.log").readlines()[-1])
sq = open(f"{O}/sdf_sq_bf16x3.txt").read()
clk = re.search(r"grad.*?clk ([0-9.]+) GHz\s+mfma_busy ([0-9.]+)", sq, re.S)
rk = {e["kernel"]: e for e in b["roofline_kernels"]}
bl = next(e for e in b["roofline_kernels"] if e["kernel"].startswith("blend"))
lat = next(e for e in b["roofline_kernels"] if "false" in e["kernel"])
oc = b["other_configs"]
# traffic from the PMC summary just written
traffic = None
for l in open("profiles/r06_bench_pmc.csv"):
    if l.startswith("sdf_mlp_split_kernel<PolBf3, true>"):
        pass
rows = list(csv.DictReader(l for l in open("profiles/r06_bench_pmc.csv") if not l.startswith("#")))
for r in rows:
    if r["kernel"] == "sdf_mlp_split_kernel<PolBf3, true>" and r["FETCH_SIZE"]:
        traffic = (2 * float(r["FETCH_SIZE"]) + float(r["WRITE_SIZE"])) * 1024 / 1e9
        busy = float(r["SQ_VALU_MFMA_BUSY_CYCLES"]) / (float(r["GRBM_GUI_ACTIVE"]) / 8 * 1024) if r.get("SQ_VALU_MFMA_BUSY_CYCLES") else None
tk = open("profiles/r06_train_kernel_stats.csv").readlines()[1]
tkern = float(re.search(r"per step: ([0-9.]+) ms", tk).group(1))
fpn = 0.0
for r in csv.DictReader(l for l in open("profiles/r06_train_kernel_stats.csv") if not l.startswith("#")):
    if re.search(r"conv3x3|deconv3x3|::wgrad_kernel|wgrad_mfma_kernel<\d+, \d+, \d, \d>|inorm|wgrad_finalize|pack_texel", r["Name"]) and "spconv" not in r["Name"]:
        fpn += float(r["MsPerStep"])
ts = b["training_step"]
f32, b16 = ts["ms_per_step"], ts["train_precision_bf16"]["ms_per_step"]
V = {
    "SHA": csrc_digest(), "HEAD_M": f"{b['value'] / 1e6:.2f}", "HEAD_MS": f"{b['ms_per_step']:.1f}",
    "SDF_MS": f"{b['kernel_ms']['sdf_mlp']:.1f}", "SDF_FRAC": f"{b['roofline']['frac']:.3f}", "SDF_TF": f"{b['roofline']['achieved']:.0f}",
    "SDF_GB": f"{traffic:.0f}" if traffic else "n/a", "MFMA_BUSY": f"{busy:.2f}" if busy else (clk.group(2) if clk else "n/a"),
    "HBM_FRAC": f"{b['roofline']['hbm_frac']:.3f}", "CLK": clk.group(1) if clk else "n/a",
    "BLEND_MS": f"{b['kernel_ms']['blend']:.1f}", "BLEND_FRAC": f"{bl['frac']:.3f}",
    "LAT_MS": f"{b['mesh_grid']['sdf_kernel_ms']:.1f}", "LAT_FRAC": f"{lat['frac']:.3f}", "MC_MS": f"{b['mesh_grid']['marching_cubes_ms']:.1f}",
    "SCENE_MS": f"{b['scene']['scene_ms']:.0f}", "BUILD_MS": f"{b['volume_build']['total_ms']:.1f}",
    "TRAIN_FP32": f"{f32:.1f}", "TRAIN_BF16": f"{b16:.1f}", "BF16_PCT": f"{100 * (f32 - b16) / f32:.1f}",
    "TRAIN_INORDER": f"{ts['in_order_ms_per_step']:.1f}", "TRAIN_FUSED": f"{ts['fused_adam_ms_per_step']:.1f}",
    "TRAIN_DDP": f"{ts['ddp_ms_per_step']:.1f}", "TRAIN_KERNEL": f"{tkern:.1f}", "FPN_MS": f"{fpn:.1f}",
    "TRAINW_FP32": " / ".join(f"{t['ms_per_step']:.1f}" for t in tr), "TRAINW_BF16": " / ".join(f"{t['ms_per_step']:.1f}" for t in tb),
    "AR_MS": f"{b['collectives']['gradient_bucket_allreduce_ms']:.3f}", "AG_MS": f"{b['collectives']['gather_rows_ms']:.2f}",
    "BAR_MS": f"{b['collectives']['barrier_ms']:.2f}",
    "TNT_M": f"{oc['tnt']['value'] / 1e6:.2f}", "SC15_M": f"{oc['scenes15']['value'] / 1e6:.2f}", "SPLIT_M": f"{oc['split_rays']['value'] / 1e6:.2f}",
    "F16_M": f"{b['other_precisions']['f16x2']['rays_per_s'] / 1e6:.2f}",
    "CPU": f"{b['cpu_baseline']['value']:.0f}", "CPU_ERR": f"{b['cpu_baseline']['max_abs_rgb_diff_vs_gpu']:.1e}", "HOSTC": str(b["cpu_baseline"]["host_cores"]),
}
for k, v in V.items():
    print(f"{k:12s} {v}")
print("split check", oc["split_rays"]["split"]["check"], "| collective_backend", b["collective_backend"])
if "--fill" in sys.argv:
    tm = open("scripts/design_tables.tmpl").read()
    parts = dict(re.findall(r"=== (\w+)\n(.*?)(?==== |\Z)", tm, re.S))
    def sub(x):
        for k, v in V.items():
            x = x.replace(f"@{k}@", v)
        assert not re.findall(r"@[A-Z0-9_]+@", x), re.findall(r"@[A-Z0-9_]+@", x)
        return x
    d = open("DESIGN.md").read()
    d = re.sub(r"(<!-- T0[^>]*-->\n).*?(<!-- /T0 -->)", lambda m: m.group(1) + sub(parts["T0"]) + m.group(2), d, flags=re.S)
    d = re.sub(r"(<!-- T5 -->\n).*?(<!-- /T5 -->)", lambda m: m.group(1) + sub(parts["T5HDR"] + parts["T5"]) + m.group(2), d, flags=re.S)
    d = re.sub(r"kernel sources `csrc_sha256 [0-9a-f]{16}`", f"kernel sources `csrc_sha256 {V['SHA']}`", d)
    open("DESIGN.md", "w").write(d)
    print("filled the T0 / T5 tables of DESIGN.md")
