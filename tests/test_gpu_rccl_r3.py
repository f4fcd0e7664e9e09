"""SURVEY 8e on hardware, as far as a 1-GPU lease goes: `bench.py` under `python -m torch.distributed.run
--nproc-per-node 1` initialises the RCCL ("nccl") process group, runs `dist.barrier()` around the timed region and the
CUDA-tensor `all_reduce` (MAX of the elapsed time, SUM of the pairs) of `sharding.reduce_timing` -- the collective code
of the 8-GPU launch, at world size 1 ("make sure the 8-GPU command cannot fail on first contact")."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_under_torchrun_one_rank_uses_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR",
                                                           "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--legs",
           "none", "--steps", "3", "--warmup", "2", "--repeats", "1", "--batch", "8"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=420, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["steps"] == 3
    assert d["collective_backend"].startswith("nccl") and "world 1" in d["collective_backend"]
