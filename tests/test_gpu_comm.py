"""kh_comm_* -- the in-library RCCL all-gather of partial MSM sums (csrc/comm.hip; SURVEY 8e, BASELINE config 4 for one process per GPU without
torch.distributed).  A 1-GPU box can only form a world of one (RCCL refuses two ranks on one device): the communicator is created, the all-gather
returns the rank's own points, and kh_msm_allreduce equals the plain MSM and the oracle.  With two or more GPUs visible the same test runs two ranks in
two processes over a shared id file, each holding one point-range shard."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import cref
from oracle import pasta as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def khip():
    import proof_systems_amd.khip as k
    k.init(0)
    return k


def test_world_of_one(khip):
    n = 1 << 12
    srs = khip.Srs.create(khip.VESTA, n)
    rnd = np.random.default_rng(3)
    F = P.Fp
    vals = [int.from_bytes(rnd.bytes(40), "little") % F.p for _ in range(n)]
    sc = cref.ints_to_limbs([F.to_mont(v) for v in vals])
    comm = khip.Comm(1, 0, khip.Comm.unique_id())
    assert khip.raw().kh_comm_world_size(comm._h) == 1 and khip.raw().kh_comm_rank(comm._h) == 0
    want_xy, want_inf = srs.msm(sc)
    got_xy, got_inf = comm.msm_allreduce(srs, sc)
    assert not got_inf and (got_xy == want_xy).all()
    g = srs.get_g()
    oxy, oinf = cref.msm(0, g, sc)                              # the oracle's C Pippenger on the same points and scalars
    assert not oinf and (np.asarray(oxy).reshape(8) == got_xy).all()
    pts = g[:5].copy(); inf = np.array([0, 1, 0, 0, 1], dtype=np.uint8)
    axy, ainf = comm.allgather_points(pts, inf)
    assert (axy == pts).all() and (ainf == inf).all()
    exy, einf = comm.allgather_points(np.zeros((0, 8), np.uint64), np.zeros(0, np.uint8))
    assert exy.shape == (0, 8)
    comm.free()


WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
import proof_systems_amd.khip as khip
rank, world, idfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
khip.init(rank)
if rank == 0:
    uid = khip.Comm.unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.rename(idfile + ".tmp", idfile)
else:
    while not os.path.exists(idfile):
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
n = 1 << 14
per = n // world
shard = khip.Srs.create(khip.VESTA, per, start=rank * per)
rng = np.random.default_rng(11)
sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); sc[:, 3] &= (1 << 60) - 1
comm = khip.Comm(world, rank, uid)
xy, inf = comm.msm_allreduce(shard, sc[rank * per:(rank + 1) * per], mont=False)
full = khip.Srs.create(khip.VESTA, n)
wxy, winf = full.msm(sc, mont=False)
assert (xy == wxy).all() and inf == bool(winf), "rank %d: sharded != whole" % rank
print("rank", rank, "ok")
"""


def test_two_ranks_when_two_devices_are_visible(khip, tmp_path):
    if khip.device_count() < 2:
        pytest.skip("one GPU visible: RCCL cannot place two ranks on one device")
    idfile = str(tmp_path / "rccl_id")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(r), "2", idfile], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("rank %d timed out" % r)
        assert p.returncode == 0 and "ok" in out, err[-2000:]
