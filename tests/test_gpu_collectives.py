"""The collectives of the N > 1 path EXECUTED on the 1-GPU box, in a world of one (RCCL refuses two ranks on one device): torch.distributed's
`nccl` (= RCCL) all_gather of device tensors through `RankShardedMsm.combine(coll_device="cuda")`, and the in-library one (kh_comm_*,
csrc/comm.hip) through the same method's `comm=` route; then `bench.py` itself with KH_BENCH_FORCE_COLLECTIVE=1, whose line must name the
collective that carried the timed loop (`config.collective_backend`) and report that the OTHER one lands on the same point
(`other_collective`).  Subprocesses with time-outs: a hung collective fails the test instead of the session."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["KH_ROOT"])
from oracle import cref
import proof_systems_amd.khip as khip
from proof_systems_amd import sharded
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
n = (1 << 13) + 7
rng = np.random.default_rng(5)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 62) - 1)
want, winf = cref.msm(0, cref.srs_generate(0, 0, n, threads=8), sc, threads=8)
eng = sharded.KhipEngine(0)
for route in ("nccl-torch", "rccl-lib"):
    comm = khip.Comm(1, 0, khip.Comm.unique_id()) if route == "rccl-lib" else None
    sm = sharded.RankShardedMsm(khip.VESTA, n, dist=dist, coll_device="cuda", engine=eng, rank=0, world=1, comm=comm, always_collective=True)
    got, ginf = sm.msm(sm.local_scalars(sc))
    assert sm.collective_backend == route, sm.collective_backend
    assert ginf == winf and np.array_equal(got, want), route
    p1 = eng.msm(sm.shard, sc); p2 = eng.msm(sm.shard, sc[::-1].copy())
    both, binf = sm.combine([p1[0], p2[0]], [p1[1], p2[1]])                 # two partials in one collective
    assert np.array_equal(both[0], p1[0]) and np.array_equal(both[1], p2[0])
    sm.close()
    if comm is not None:
        comm.free()
# without always_collective a lone rank skips the collective
sm = sharded.RankShardedMsm(khip.VESTA, 64, dist=dist, coll_device="cuda", engine=eng, rank=0, world=1)
sm.msm(sc[:64]); assert sm.collective_backend is None
dist.destroy_process_group()
print("COLLECTIVES_OK")
"""


def _env(**kw):
    return dict(os.environ, KH_ROOT=ROOT, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **kw)


def test_world_of_one_nccl_and_library_collectives(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    p = subprocess.run([sys.executable, str(script)], env=_env(MASTER_PORT="29713"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0 and b"COLLECTIVES_OK" in p.stdout, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("comm,timed,other", [("torch", "nccl-torch", "rccl-lib"), ("lib", "rccl-lib", "nccl-torch")])
def test_bench_line_names_the_collective_that_ran(comm, timed, other):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--log-n", "14", "--no-oplist"]
    p = subprocess.run(cmd, env=_env(KH_BENCH_FORCE_COLLECTIVE="1", KH_BENCH_COMM=comm, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_PORT="29717"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    line = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["config"]["collective_backend"] == timed and line["config"]["process_group_backend"] == "nccl"
    oc = line["other_collective"]
    assert oc.get("error") is None and oc["ran"] == other and oc["same_point_as_timed_collective"] is True, oc
    assert line["cpu_baseline"]["gpu_result_matches"] is True
