"""N>1 paths with world_size 2 over gloo.

CPU (collected by `-m "not gpu"`): `proof_systems_amd.sharded.RankShardedMsm` -- the product's sharding, packing,
all-gather and fold code (the fold is the library's own kh_points_sum, host code that needs no GPU) -- with the ORACLE as
the per-rank compute engine, because this container has no GPU: what is tested is the sharding arithmetic and the
collective, fold(all_gather(partials)) == MSM over the whole range.
GPU (`-m gpu`): the same workers with the product engine (libkimchi_hip: kh_srs_create_device_range per rank, kh_msm, the
fold), two gloo ranks sharing the one GPU of the test box, against one oracle MSM over the whole range; and the
coset-sharded d8 extension through kh_coset_ntt_dev against the single-GPU kh_lde; and the one-process sharding
(`LocalShardedMsm`: a handle per shard, a host thread per shard, per-device contexts)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["KH_ROOT"])
from oracle import cref
from proof_systems_amd import sharded
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
PRODUCT = os.environ.get("KH_ENGINE") == "product"
n_total = int(os.environ.get("KH_TOTAL", "1024")) + 3            # ragged: shard sizes differ by one

class OracleEngine:                                              # CPU stand-in for the GPU (no GPU in the CPU test tier)
    def __init__(self):
        import proof_systems_amd.khip as khip                    # loads without a GPU; only kh_points_sum (host code) is called
        self.khip = khip
    def make_shard(self, curve, start, count): return cref.srs_generate(curve, start, count)
    def msm(self, shard, scalars, mont=True):
        xy, inf = cref.msm(0, shard, scalars, scalars_mont=mont)
        return xy, inf
    def points_sum(self, curve, xy, inf):
        out, oinf = self.khip.points_sum(curve, xy, inf)         # the product's fold
        return out, bool(oinf)
    def free_shard(self, shard): pass

engine = sharded.KhipEngine(0) if PRODUCT else OracleEngine()
sm = sharded.RankShardedMsm(0, n_total, dist=dist, coll_device="cpu", engine=engine, rank=rank, world=world)
rng = np.random.default_rng(1234)                                # every rank draws the same full vector, uses its slice
sc = rng.integers(0, 1 << 64, size=(n_total, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 62) - 1)
got, ginf = sm.msm(sm.local_scalars(sc))
# two MSMs of a phase in ONE collective
p1 = engine.msm(sm.shard, sm.local_scalars(sc)); p2 = engine.msm(sm.shard, sm.local_scalars(sc[::-1].copy()))
both, binf = sm.combine([p1[0], p2[0]], [p1[1], p2[1]])
gs = cref.srs_generate(0, 0, n_total, threads=8)
want, winf = cref.msm(0, gs, sc, threads=8)
assert ginf == winf and np.array_equal(got, want), "sharded MSM fold mismatch"
assert bool(binf[0]) == winf and np.array_equal(both[0], want)
s, c = sharded.shard_range(n_total, world, rank)
want2, winf2 = cref.msm(0, gs, np.concatenate([sc[::-1][sharded.shard_range(n_total, world, r)[0]:][:sharded.shard_range(n_total, world, r)[1]] for r in range(world)]), threads=8)
assert bool(binf[1]) == winf2 and np.array_equal(both[1], want2)
# the `comm=` route of combine (the in-library collective's shape: kh_comm_allgather_points returns [world * k] points, rank-major), with a
# stand-in that moves the same records over gloo -- RCCL needs one GPU per rank, which no test box has
class WireComm:
    def __init__(self, world): self.world = world
    def allgather_points(self, xy, inf):
        rec = [None] * self.world
        dist.all_gather_object(rec, (np.asarray(xy).copy(), np.asarray(inf).copy()))
        return np.concatenate([r[0].reshape(-1, 8) for r in rec]), np.concatenate([r[1].reshape(-1) for r in rec])
sm2 = sharded.RankShardedMsm.__new__(sharded.RankShardedMsm)
sm2.__dict__.update(sm.__dict__); sm2.comm = WireComm(world); sm2.dist = None; sm2.collective_backend = None
both2, binf2 = sm2.combine([p1[0], p2[0]], [p1[1], p2[1]])
assert np.array_equal(both2, both) and np.array_equal(binf2, binf) and sm2.collective_backend == "rccl-lib" and sm.collective_backend == "gloo-torch"
sm.close()
dist.barrier()
if rank == 0:
    print("GLOO_OK")
dist.destroy_process_group()
"""


def _run_world2(tmp_path, text, port, extra_env=None, token="GLOO_OK"):
    script = tmp_path / "worker.py"
    script.write_text(text)
    env = dict(os.environ, KH_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert token in outs[0]


def test_point_range_sharding_world2(tmp_path):
    _run_world2(tmp_path, WORKER, 29617)


@pytest.mark.gpu
def test_point_range_sharding_world2_product(tmp_path):
    """The shipped path across ranks: kh_srs_create_device_range slices, kh_msm, all-gather, kh_points_sum (two gloo ranks
    on the one GPU of the box; the driver's multi-GPU runs use the same code over RCCL with one GPU per rank)."""
    _run_world2(tmp_path, WORKER, 29621, {"KH_ENGINE": "product", "KH_TOTAL": str(1 << 17)})


@pytest.mark.gpu
def test_one_process_sharding_product():
    """LocalShardedMsm: ONE process, a handle per shard created through kh_set_device + kh_srs_create_device_range, a host
    thread per shard (ctypes releases the GIL), the fold -- over every visible device, and with 3 shards on device 0 so that
    the path is exercised on a one-GPU box too."""
    from oracle import cref
    import proof_systems_amd.khip as khip
    from proof_systems_amd import sharded
    khip.init(0)
    n = (1 << 16) + 5
    rng = np.random.default_rng(99)
    sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 62) - 1)
    want, winf = cref.msm(1, cref.srs_generate(1, 0, n, threads=8), sc, threads=8)
    for devices in ([0, 0, 0], list(range(khip.device_count()))):
        sm = sharded.LocalShardedMsm(khip.PALLAS, n, devices)
        assert [khip._lib.kh_srs_device(s._h) for s in sm.shards] == devices
        got, ginf = sm.msm(sc)
        sm.close()
        assert bool(ginf) == winf and np.array_equal(got, want), devices
    assert khip.get_device() == 0


WORKER_COSETS = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["KH_ROOT"])
from oracle import cref
from oracle import pasta as P
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
F = P.Fp; logn = 6; n = 1 << logn; cols = 3
rng = np.random.default_rng(7)                              # every rank holds the (small) coefficient vectors
coeffs = [[int.from_bytes(rng.bytes(40), "little") % F.p for _ in range(n)] for _ in range(cols)]
om8 = F.root_of_unity(logn + 3)
mine = [r for r in range(8) if r % world == rank]           # cosets of this rank (one per GPU at world = 8)
local = {}
for r in mine:                                              # coset r = NTT_n(c_j * w_{8n}^{r j}): what kh_coset_ntt_dev computes
    sh = pow(om8, r, F.p)
    local[r] = [P.ntt(F, [c * pow(sh, j, F.p) % F.p for j, c in enumerate(col)], logn) for col in coeffs]
# a row-wise step that needs the NEXT row (z(x w)): stays inside the coset
local_next = {r: [[col[(i + 1) % n] for i in range(n)] for col in local[r]] for r in mine}
# only the final 8n-point vector needs an exchange: all-gather the cosets, interleave
flat = np.array([[F.to_mont(v) for col in local[r] for v in col] for r in mine], dtype=object)
t = torch.from_numpy(cref.ints_to_limbs([int(v) for v in flat.reshape(-1)]).view(np.int64).copy())
allt = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(allt, t)
full = [[None] * (8 * n) for _ in range(cols)]
nxt = [[None] * (8 * n) for _ in range(cols)]
for rk in range(world):
    vals = [F.from_mont(v) for v in cref.limbs_to_ints(allt[rk].numpy().view(np.uint64))]
    rs = [r for r in range(8) if r % world == rk]
    for k, r in enumerate(rs):
        for c in range(cols):
            seg = vals[(k * cols + c) * n:(k * cols + c + 1) * n]
            for i in range(n):
                full[c][8 * i + r] = seg[i]
for c in range(cols):
    want = P.lde(F, coeffs[c], logn, 3)
    assert full[c] == want, "interleaved cosets != d8 extension"
    for r in mine:                                          # next row of the d8 vector (shift by 8) == next element of the coset
        assert local_next[r][c] == [want[(8 * (i + 1) + r) % (8 * n)] for i in range(n)]
dist.barrier()
if rank == 0:
    print("GLOO_COSETS_OK")
dist.destroy_process_group()
"""


def test_coset_sharded_lde_world2(tmp_path):
    """SURVEY 8e, NTT side: the d8 extension sharded by coset (rank r computes the cosets r mod world of every column);
    row-wise steps incl. the next-row access stay rank-local, one all-gather rebuilds the interleaved 8n vector."""
    _run_world2(tmp_path, WORKER_COSETS, 29619, token="GLOO_COSETS_OK")


WORKER_COSETS_PRODUCT = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["KH_ROOT"])
import proof_systems_amd.khip as khip
from proof_systems_amd import sharded
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
khip.init(0)
logn, cols = 12, 5; n = 1 << logn
rng = np.random.default_rng(7)
coeffs = rng.integers(0, 1 << 64, size=(cols, n, 4), dtype=np.uint64); coeffs[:, :, 3] &= np.uint64((1 << 61) - 1)
om8 = np.array(khip.domain_generator(khip.FP, logn + 3), dtype=np.uint64).reshape(1, 4)
one = np.array([0x34786d38fffffffd, 0x992c350be41914ad, 0xffffffffffffffff, 0x3fffffffffffffff], dtype=np.uint64).reshape(1, 4)
src = khip.DevBuf(coeffs.nbytes).upload(coeffs)
mine = sharded.coset_shard_ids(world, rank)
local = np.zeros((len(mine), cols, n, 4), dtype=np.uint64)
dst = khip.DevBuf(coeffs.nbytes)
for k, r in enumerate(mine):
    sh = one.copy()
    for _ in range(r):
        sh = khip.debug_field_op(khip.FP, "mul", sh, om8)           # w_8n^r
    khip.coset_ntt_dev(khip.FP, src, logn, sh[0], dst, cols)
    local[k] = dst.download((cols, n, 4))
t = torch.from_numpy(local.view(np.int64).copy())
allt = [torch.empty_like(t) for _ in range(world)]
dist.all_gather(allt, t)
full = sharded.interleave_cosets([a.numpy().view(np.uint64) for a in allt], world, n, cols)
want = khip.lde(khip.FP, coeffs, logn, 3)                           # the single-GPU d8 extension (parity-tested against the oracle elsewhere)
assert np.array_equal(full, want), "interleaved cosets != kh_lde"
dist.barrier()
if rank == 0:
    print("GLOO_COSETS_OK")
dist.destroy_process_group()
"""


@pytest.mark.gpu
def test_coset_sharded_lde_world2_product(tmp_path):
    _run_world2(tmp_path, WORKER_COSETS_PRODUCT, 29623, token="GLOO_COSETS_OK")
