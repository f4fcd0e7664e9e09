"""N>1 path on CPU: two `gloo` ranks shard independent streams (no data-path collective), run the
front-end restatement (oracle, as the stand-in for the device path) on their own streams, and
reduce timing/unit counts exactly as bench.py does.  Checks: shards are disjoint and cover all
streams; a stream's output does not depend on the rank that processed it; MAX/SUM reduction."""
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_streams(stream_ids, n_frames=3):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from kimera_vio_amd import params as P
    L = P.load_camera_params(os.path.join(G, "sensorLeft.yaml"))
    R = P.load_camera_params(os.path.join(G, "sensorRight.yaml"))
    p = P.load_frontend_params(os.path.join(G, "params_euroc", "FrontendParams.yaml"), use_ransac=0)
    p.detector.max_features_per_frame = 100
    z = np.load(os.path.join(G, "micro_euroc_f10_18.npz"))
    out = {}
    for s in stream_ids:
        fe = O.Frontend(L, R, p)
        sig = []
        for i in range(n_frames):
            j = (s + i) % len(z["lefts"])  # every stream starts at a different frame
            o = fe.process(z["lefts"][j], z["rights"][j], int(z["timestamps"][0]) + i * 50_000_000)
            sig.append((o["n_keypoints"], float(np.asarray(o["keypoints"], np.float64).sum()),
                        int(np.asarray(o["landmarks"]).sum())))
        out[s] = sig
    return out


def _worker(rank, world, port, n_streams, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import time

    import torch.distributed as dist
    from kimera_vio_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, _, w = sharding.env_rank()
        mine = sharding.shard_streams(n_streams, w, r)
        sharding.barrier(dist, w)
        t0 = time.perf_counter()
        res = _run_streams(mine)
        sharding.barrier(dist, w)
        el = time.perf_counter() - t0
        units = sum(len(v) for v in res.values())
        el_max, units_sum = sharding.reduce_timing(dist, w, el, units)
        q.put((rank, mine, res, el, el_max, units_sum))
    finally:
        dist.destroy_process_group()


def test_shard_streams_partition():
    sys.path.insert(0, ROOT)
    from kimera_vio_amd import sharding
    for n, w in ((8, 8), (64, 8), (5, 2), (3, 4), (1, 1)):
        shards = [sharding.shard_streams(n, w, r) for r in range(w)]
        flat = sorted(s for sh in shards for s in sh)
        assert flat == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    with pytest.raises(ValueError):
        sharding.shard_streams(4, 2, 2)


def test_two_gloo_ranks_shard_streams_and_reduce_timing():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world, n_streams = 2, 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_streams, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda g: g[0])
    all_streams = sorted(s for g in got for s in g[1])
    assert all_streams == list(range(n_streams))
    assert got[0][4] == got[1][4] == max(got[0][3], got[1][3])        # MAX over ranks
    assert got[0][5] == got[1][5] == n_streams * 3                    # SUM of stereo pairs
    # a stream's result is independent of where it ran
    single = _run_streams(range(n_streams))
    for g in got:
        for s, sig in g[2].items():
            assert sig == single[s]


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` without a torchrun environment re-executes itself under
    `python -m torch.distributed.run --nproc-per-node 2` (the flag used to be ignored) and, with
    --config c4, shards BASELINE configs[3]'s 8 sequences over the ranks (sequence q on rank q mod world).
    --dry-run = launch plumbing over gloo only: no GPU, nothing computed."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                           "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "c4",
                        "--steps", "5", "--dry-run"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert [x["sequences"] for x in d["ranks"]] == [[0, 2, 4, 6], [1, 3, 5, 7]]
    assert d["pairs_sum"] == 8 * 5 and d["elapsed_max"] == 2.0          # SUM of units, MAX of rank times
    assert d["ranks"][0]["first_pixel_sum"] != d["ranks"][1]["first_pixel_sum"]   # different sequences


def test_bench_eight_ranks_dry_run_headline_config():
    """the driver's 8-GPU launch line on BASELINE configs[2] (c3, weak scaling: every rank its
    own streams) and on configs[3] (c4: one EuRoC sequence per GPU) -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py --gpus 8` with --dry-run: 8 gloo ranks build their shards,
    meet at the barrier and reduce the timing; no GPU, nothing computed."""
    import json
    import socket
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                           "MASTER_PORT")}
    for config, steps in (("c3", 3), ("c4", 2)):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                            "--gpus", "8", "--config", config, "--steps", str(steps), "--warmup", "1", "--dry-run"],
                           capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert d["n_gpus"] == 8 and len(d["ranks"]) == 8 and [x["rank"] for x in d["ranks"]] == list(range(8))
        assert d["elapsed_max"] == 8.0                       # MAX over the ranks' times (rank r reports 1 + r)
        if config == "c3":
            assert d["scaling"] == "weak" and d["pairs_sum"] == sum(x["batch"] for x in d["ranks"]) * steps
            assert len({x["first_pixel_sum"] for x in d["ranks"]}) == 8   # every rank renders its own streams
        else:
            assert d["scaling"] == "strong" and [x["sequences"] for x in d["ranks"]] == [[q] for q in range(8)]
            assert d["pairs_sum"] == 8 * steps
