"""N > 1 path on CPU: two gloo processes shard 5 scenes, time a fake step, MAX-reduce and gather the records."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from surf_amd import dist as D
    rank, local_rank, world = D.init_from_env(backend="gloo")
    scenes = D.shard_scenes(5, rank, world)
    D.barrier()
    t0 = time.perf_counter(); time.sleep(0.05 * (rank + 1)); dt = time.perf_counter() - t0
    D.barrier()
    tmax = D.max_over_ranks(dt)
    recs = D.gather_records({"rank": rank, "scenes": scenes, "dt": dt})
    if rank == 0:
        print(json.dumps({"world": world, "tmax": tmax, "recs": recs}))
    D.shutdown()
""") % ROOT


def test_two_rank_gloo_sharding_and_timing(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = json.loads(outs[0][0].strip().splitlines()[-1])
    assert res["world"] == 2
    assert sorted(sum((r["scenes"] for r in res["recs"]), [])) == [0, 1, 2, 3, 4]
    assert res["recs"][0]["scenes"] == [0, 2, 4] and res["recs"][1]["scenes"] == [1, 3]
    assert res["tmax"] >= max(r["dt"] for r in res["recs"]) - 1e-9 and res["tmax"] >= 0.09


GATHER_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from surf_amd import dist as D
    rank, local_rank, world = D.init_from_env(backend="gloo")
    R = 11                                              # an image of 11 rays split 5 + 6 (bench.py --split rays: r R / N .. (r + 1) R / N)
    r0, r1 = rank * R // world, (rank + 1) * R // world
    mine = torch.arange(R * 3, dtype=torch.float32).reshape(R, 3)[r0:r1] * 0.5
    whole = D.gather_rows(mine)
    if rank == 0:
        print(json.dumps({"rows": whole.shape[0], "equal": bool(torch.equal(whole, torch.arange(R * 3, dtype=torch.float32).reshape(R, 3) * 0.5))}))
    else:
        assert whole is None
    D.shutdown()
""") % ROOT


def test_two_rank_gloo_gather_rows_stitches_uneven_shares(tmp_path):
    """The single-scene split's data-path collective (surf_amd.dist.gather_rows) over gloo, world_size 2, uneven row counts."""
    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    assert json.loads(outs[0][0].strip().splitlines()[-1]) == {"rows": 11, "equal": True}


def test_single_process_helpers():
    from surf_amd import dist as D
    import torch
    t = torch.arange(6.0).reshape(2, 3)
    assert D.gather_rows(t) is t
    assert D.shard_scenes(15, 3, 8) == [3, 11]
    assert D.max_over_ranks(1.5) == 1.5
    assert D.gather_records({"a": 1}) == [{"a": 1}]


# ---- bench.py's own N > 1 control flow (kernels stubbed by --dry, gloo instead of RCCL) --------------------------------
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return env


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the parent spawns two ranks, rank 0's line says n_gpus 2 and the 5
    scenes are dealt round-robin (datasets/__init__.py:37-38)."""
    import json
    res = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1", "--scenes", "5"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["data"] == "dry-run"
    assert [(x["scene"], x["rank"]) for x in r["scenes"]] == [(0, 0), (1, 1), (2, 0), (3, 1), (4, 0)]
    assert r["config"]["rays_per_step"] == 5 * 576 * 800
    # whole-job value = all rays of all ranks / the slowest rank's time; rank 0 renders 3 scenes of >= 2 ms per step
    assert r["ms_per_step"] >= 6.0
    assert abs(r["value"] - r["config"]["rays_per_step"] / (r["ms_per_step"] * 1e-3)) < 1e-6 * r["value"]


def test_bench_under_a_launcher_and_world_size_mismatch():
    """The torch.distributed.run contract: ranks come from the environment; a --gpus that disagrees with WORLD_SIZE is
    an error, not a silent single-rank run."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, BENCH, "--gpus", "2", "--dry", "--steps", "2", "--warmup", "0"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    r = json.loads([l for l in outs[0][0].splitlines() if l.startswith("{")][-1])
    assert r["n_gpus"] == 2 and [x["scene"] for x in r["scenes"]] == [0, 1]
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]      # only rank 0 prints the line
    bad = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], env=dict(_clean_env(), WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode == 2 and "WORLD_SIZE" in bad.stderr


GRAD_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from surf_amd import dist as D
    rank, local_rank, world = D.init_from_env(backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(300, 7)), torch.nn.Parameter(torch.zeros(11)), torch.nn.Parameter(torch.zeros(4, 4))]
    params[0].grad = torch.full((300, 7), float(rank + 1))
    params[1].grad = torch.arange(11.0) * (rank + 1)
    # params[2] has no gradient on rank 1 (e.g. a volume level no ray touched)
    if rank == 0:
        params[2].grad = torch.ones(4, 4)
    n = D.all_reduce_gradients(params, bucket_bytes=4096)
    if rank == 0:
        print(json.dumps({"buckets": n, "g0": float(params[0].grad.mean()), "g1": params[1].grad.tolist(), "g2": float(params[2].grad.mean())}))
    else:
        assert abs(float(params[2].grad.mean()) - 0.5) < 1e-6
    D.shutdown()
""") % ROOT


def test_gradient_all_reduce_two_ranks(tmp_path):
    """dist.all_reduce_gradients (the DDP averaging of runner.py:102 done explicitly): bucketing, missing gradients, mean."""
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    res = json.loads([l for l in outs[0][0].strip().splitlines() if l.startswith("{")][-1])
    assert res["buckets"] == 2                       # [8400 B] (alone: it exceeds the 4096-byte bucket), [44 B + 64 B]
    assert abs(res["g0"] - 1.5) < 1e-6 and abs(res["g2"] - 0.5) < 1e-6
    assert all(abs(a - 1.5 * k) < 1e-6 for k, a in enumerate(res["g1"]))


def test_bench_train_workload_two_ranks_ddp_dry():
    """`python bench.py --workload train --gpus 2` (BASELINE configs[3]): two fresh ranks, the model wrapped in
    DistributedDataParallel (gloo here, RCCL on GPUs), the runner's step sequence, one line with training rays/s, steps/s
    and the all-reduce time of the gradient bucket."""
    import json
    res = subprocess.run([sys.executable, BENCH, "--workload", "train", "--gpus", "2", "--dry", "--steps", "3", "--warmup", "1"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak" and r["data"] == "dry-run"
    assert r["config"]["parallelism"] == "ddp2" and r["config"]["rays_per_rank_step"] == 512
    assert r["gradient_allreduce_ms"] > 0 and r["config"]["trainable_parameters"] == 1410000
    assert abs(r["value"] - 2 * 512 * r["steps_per_s"]) < 1e-6 * r["value"]


def test_bench_kills_the_surviving_ranks_when_one_dies():
    """A rank that exits before the first barrier must not leave the others waiting in a collective until the caller's own
    limit: the launcher kills exactly the processes it started, reports the exit codes and returns non-zero - quickly."""
    import time
    for workload in ("dtu", "train"):
        t0 = time.monotonic()
        res = subprocess.run([sys.executable, BENCH, "--workload", workload, "--gpus", "2", "--dry", "--steps", "2", "--warmup", "0",
                              "--fail-rank", "1"], env=_clean_env(), capture_output=True, text=True, timeout=120)
        assert res.returncode == 1, (workload, res.returncode, res.stderr[-1000:])
        assert "rank exit codes" in res.stderr and "killed" in res.stderr
        assert time.monotonic() - t0 < 60


STATE_WORKER = textwrap.dedent("""
    import json, os, sys
    sys.path.insert(0, %r)
    import torch
    from surf_amd import dist as D
    rank, local_rank, world = D.init_from_env(backend="gloo")
    torch.manual_seed(100 + rank)                       # replicas that start DIFFERENT (another seed, or a checkpoint loaded on rank 0 only)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 2))
    net(torch.randn(16, 5))                             # moves the BatchNorm running statistics, differently per rank
    before = float(sum(p.detach().abs().sum() for p in net.parameters()) + sum(b.float().abs().sum() for b in net.buffers()))
    versions = [t._version for t in list(net.parameters()) + list(net.buffers())]
    n_all = D.broadcast_module_state(net, src=0)
    # every weight cache of the package is keyed on (parameter._version, data_ptr): the broadcast must be visible there
    versions_bumped = all(t._version > v for t, v in zip(list(net.parameters()) + list(net.buffers()), versions))
    after = float(sum(p.detach().abs().sum() for p in net.parameters()) + sum(b.float().abs().sum() for b in net.buffers()))
    with torch.no_grad():
        net[1].running_mean.add_(float(rank + 1))       # per-rank drift of the buffers during training
    n_buf = D.broadcast_module_state(net, src=0, buffers_only=True)
    rm = float(net[1].running_mean.sum())
    # a cache of the package itself: inv_s read before the broadcast (a warm-up / val forward) must be re-read after it
    from surf_amd.implicit_surface import SingleVarianceNetwork
    dev_net = SingleVarianceNetwork(0.1 * (rank + 1))
    stale = dev_net.inv_s()
    D.broadcast_module_state(dev_net, src=0)
    inv_s = dev_net.inv_s()
    print("RESULT " + json.dumps({"rank": rank, "before": before, "after": after, "n_all": n_all, "n_buf": n_buf, "rm": rm,
                                  "versions_bumped": versions_bumped, "inv_s": inv_s, "stale": stale}))
    D.shutdown()
""") % ROOT


def test_broadcast_module_state_makes_replicas_identical(tmp_path):
    """dist.broadcast_module_state = what DistributedDataParallel does at wrap time (parameters + buffers from rank 0) and at
    every forward (buffers), for a model that is not wrapped: ranks started from different seeds hold rank 0's state afterwards,
    BatchNorm running statistics included."""
    import json
    script = tmp_path / "state_worker.py"
    script.write_text(STATE_WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [subprocess.Popen([sys.executable, str(script)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                              env=dict(_clean_env(), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                       MASTER_PORT=str(port))) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    res = sorted((json.loads([ln for ln in o[0].splitlines() if ln.startswith("RESULT ")][-1][7:]) for o in outs), key=lambda r: r["rank"])
    assert res[0]["before"] != res[1]["before"]
    assert res[0]["after"] == res[1]["after"] == res[0]["before"]
    assert res[0]["n_all"] == res[1]["n_all"] == 6 + 3 and res[0]["n_buf"] == 3      # 6 parameters + 3 BatchNorm buffers
    assert res[0]["rm"] == res[1]["rm"]
    assert res[0]["versions_bumped"] and res[1]["versions_bumped"]
    assert res[0]["stale"] != res[1]["stale"] and res[1]["inv_s"] == res[0]["inv_s"] == res[0]["stale"]
