"""N>1 path on CPU (gloo, world_size 2): point-range sharding of an MSM + all-gather of the
partial sums + local fold, exactly the data flow bench.py runs over RCCL.  The
per-rank MSM here is the oracle's (no GPU in this container); what is tested is the sharding
arithmetic and the collective: fold(all_gather(partials)) == MSM over the whole range."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["KH_ROOT"])
from oracle import cref
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
n = 512
g = cref.srs_generate(0, rank * n, n)                      # this rank's slice of the bases
rng = np.random.default_rng(1234 + rank)
sc = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); sc[:, 3] &= np.uint64((1 << 62) - 1)
part, pinf = cref.msm(0, g, sc)
mine = torch.from_numpy(np.concatenate([part, np.array([int(pinf)], dtype=np.uint64)]).view(np.int64).copy())
allp = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(allp, mine)
parts = torch.stack(allp).numpy().view(np.uint64)
acc, ainf = parts[0, :8].copy(), bool(parts[0, 8])
for r in range(1, world):
    acc, ainf = cref.point_add(0, acc, parts[r, :8].copy(), ainf, bool(parts[r, 8]))
# reference: one MSM over the whole range with the same per-rank scalars
gs = cref.srs_generate(0, 0, world * n)
scs = []
for r in range(world):
    s = np.random.default_rng(1234 + r).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 62) - 1)
    scs.append(s)
want, winf = cref.msm(0, gs, np.concatenate(scs))
assert ainf == winf and np.array_equal(acc, want), "sharded MSM fold mismatch"
dist.barrier()
if rank == 0:
    print("GLOO_OK")
dist.destroy_process_group()
"""


def test_point_range_sharding_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, KH_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_OK" in outs[0]


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
    script = tmp_path / "worker_cosets.py"
    script.write_text(WORKER_COSETS)
    env = dict(os.environ, KH_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29619", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GLOO_COSETS_OK" in outs[0]
