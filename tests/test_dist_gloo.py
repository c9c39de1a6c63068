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


def test_single_process_helpers():
    from surf_amd import dist as D
    assert D.shard_scenes(15, 3, 8) == [3, 11]
    assert D.max_over_ranks(1.5) == 1.5
    assert D.gather_records({"a": 1}) == [{"a": 1}]
