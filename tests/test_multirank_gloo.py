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
