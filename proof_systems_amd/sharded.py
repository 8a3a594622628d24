"""Multi-GPU sharding of the hot path above the C ABI.

MSM: the POINT RANGE is partitioned -- shard r owns the bases g[r*n/R, (r+1)*n/R) (expanded to window tables once, at
SRS upload) and the matching scalars, runs the complete single-GPU Pippenger and emits ONE partial sum.  Group addition
is not a reduction operator RCCL offers, so the combine is an all-gather of the R partials (72 bytes each) followed by a
local fold (`kh_points_sum`, host code of the library) on every rank; partials of several MSMs of a phase travel in one
collective.  There is no other data-path communication.

Two deployments share this file:
  * `RankShardedMsm`  -- one process per GPU under torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" lets
                         several ranks share one GPU for tests).  What `bench.py --gpus N` times.
  * `LocalShardedMsm` -- ONE process, one SRS handle per device (`kh_set_device` + `kh_srs_create_device_range`): with the
                         product engine the whole MSM is ONE library call, `kh_msm_sharded` (native thread per shard, fold in
                         the library) -- what a single Rust prover process calls (`GpuShardedMsm` in rust/kimchi-hip);
                         BASELINE config 4 without torchrun (config 5 = one curve per device: bench.py --pair).

The compute engine is a parameter so that the CPU-only tests can run the very same sharding / collective / fold code
with the oracle standing in for the GPU (tests/test_multirank_gloo.py); the product never does that: the default
engine is `KhipEngine`, which fails loudly without the HIP library and a GPU.

LDE: the d8 extension shards by COSET -- shard r evaluates every column on w_8n^r <w_n> (`kh_coset_ntt_dev`), every
row-wise step of the quotient incl. the next-row access stays shard-local, one all-gather rebuilds the interleaved
8n vector before the final iNTT (`coset_shard_ids`, `interleave_cosets`)."""
from __future__ import annotations

import threading

import numpy as np


class KhipEngine:
    """The product: libkimchi_hip.so through proof_systems_amd.khip."""

    def __init__(self, device: int = -1):
        import proof_systems_amd.khip as khip
        self.khip = khip
        khip.init(device)

    def make_shard(self, curve: int, start: int, count: int):
        return self.khip.Srs.create(curve, count, start=start)          # SRS::create on the device + window tables

    def msm(self, shard, scalars, mont: bool = True):
        xy, inf = shard.msm(scalars, mont=mont)
        return xy, bool(inf)

    def msm_dev(self, shard, scalars_dev_ptr: int, n: int, mont: bool = True):
        xy, inf = shard.msm_batch_dev(scalars_dev_ptr, n, 1, mont=mont)
        return xy[0], bool(inf[0])

    def points_sum(self, curve: int, xy, inf):
        out, oinf = self.khip.points_sum(curve, xy, inf)
        return out, bool(oinf)

    def free_shard(self, shard):
        shard.close()


def shard_range(total: int, world: int, rank: int):
    """[start, start + count) of shard `rank`: contiguous, sizes differ by at most one."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def pack_partials(xy_list, inf_list):
    """k partial sums -> one int64 tensor payload [k, 9]: 8 limbs + the infinity flag."""
    k = len(xy_list)
    buf = np.zeros((k, 9), dtype=np.uint64)
    for j in range(k):
        buf[j, :8] = xy_list[j]
        buf[j, 8] = 1 if inf_list[j] else 0
    return buf


class RankShardedMsm:
    """One process per GPU.  `dist` is an initialised torch.distributed module (or None for a single rank)."""

    def __init__(self, curve: int, total_points: int, dist=None, coll_device: str = "cpu", engine=None, rank: int = 0, world: int = 1, comm=None,
                 always_collective: bool = False):
        """comm: a proof_systems_amd.khip.Comm (kh_comm_*: the IN-LIBRARY RCCL all-gather, what a Rust / C caller of kh_msm_allreduce gets) -- the
        combine then never touches torch.distributed.  always_collective: run the collective even in a world of one (how a 1-GPU box exercises
        the RCCL call path; by default a lone rank skips it)."""
        self.curve, self.total, self.dist, self.coll_device = curve, total_points, dist, coll_device
        self.rank, self.world = rank, world
        self.comm, self.always_collective = comm, always_collective
        self.collective_backend = None                      # set by the first combine that ran a collective: "rccl-lib" | "nccl-torch" | "gloo-torch"
        self.engine = engine if engine is not None else KhipEngine()
        self.start, self.count = shard_range(total_points, world, rank)
        self.shard = self.engine.make_shard(curve, self.start, self.count)

    def local_scalars(self, scalars):
        """This rank's slice of a full-length scalar vector."""
        return scalars[self.start:self.start + self.count]

    def combine(self, partial_xy, partial_inf):
        """All-gather the per-rank partial sums of k MSMs ([k, 8] limbs, [k] flags) and fold them on every rank."""
        xy = np.asarray(partial_xy, dtype=np.uint64).reshape(-1, 8)
        inf = np.asarray(partial_inf).reshape(-1)
        k = xy.shape[0]
        if (self.dist is None and self.comm is None) or (self.world == 1 and not self.always_collective):
            return xy, inf.astype(bool)
        if self.comm is not None:                                            # csrc/comm.hip: ncclAllGather on the communicator's own stream
            axy, ainf = self.comm.allgather_points(xy, inf.astype(np.uint8))
            parts = np.concatenate([np.asarray(axy, dtype=np.uint64).reshape(self.world, k, 8),
                                    np.asarray(ainf, dtype=np.uint64).reshape(self.world, k, 1)], axis=2)
            self.collective_backend = "rccl-lib"
        else:
            import torch
            mine = torch.from_numpy(pack_partials(list(xy), list(inf)).view(np.int64)).to(self.coll_device)
            allp = [torch.empty_like(mine) for _ in range(self.world)]
            self.dist.all_gather(allp, mine)
            parts = torch.stack(allp).cpu().numpy().view(np.uint64)        # [world, k, 9]
            self.collective_backend = ("nccl" if self.coll_device == "cuda" else "gloo") + "-torch"
        out = np.zeros((k, 8), dtype=np.uint64); oinf = np.zeros(k, dtype=bool)
        for j in range(k):
            out[j], oinf[j] = self.engine.points_sum(self.curve, parts[:, j, :8].copy(), parts[:, j, 8].astype(np.uint8))
        return out, oinf

    def msm(self, local_scalars, mont: bool = True):
        """The whole MSM: local slice through the single-GPU pipeline, then the combine.  Every rank gets the result."""
        xy, inf = self.engine.msm(self.shard, local_scalars, mont=mont)
        out, oinf = self.combine([xy], [inf])
        return out[0], bool(oinf[0])

    def close(self):
        self.engine.free_shard(self.shard)


class LocalShardedMsm:
    """One process, several devices (or several shards on one device): a handle per shard, a host thread per shard."""

    def __init__(self, curve: int, total_points: int, devices, engine_factory=None):
        self.curve, self.total = curve, total_points
        self.devices = list(devices)
        R = len(self.devices)
        self.ranges = [shard_range(total_points, R, r) for r in range(R)]
        self.engines, self.shards = [], []
        for r, dev in enumerate(self.devices):
            eng = engine_factory(dev) if engine_factory else KhipEngine(dev)
            if hasattr(eng, "khip"):
                eng.khip.set_device(dev)                      # the handle lives on the device current at creation
            self.engines.append(eng)
            self.shards.append(eng.make_shard(curve, *self.ranges[r]))

    def msm(self, scalars, mont: bool = True):
        R = len(self.shards)
        if all(isinstance(e, KhipEngine) for e in self.engines):
            # the product: ONE library call (kh_msm_sharded: the slices go to their devices from native threads, the fold is the
            # library's) -- what the Rust shim's GpuShardedMsm calls; no Python thread or GIL in the path
            sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
            assert [self.ranges[r][0] for r in range(R)] == [sum(c for _, c in self.ranges[:r]) for r in range(R)]
            return self.engines[0].khip.msm_sharded(self.shards, sc, mont=mont)
        parts = [None] * R
        errs = []

        def work(r):
            try:
                s, c = self.ranges[r]
                parts[r] = self.engines[r].msm(self.shards[r], scalars[s:s + c], mont=mont)     # runs on the handle's device
            except Exception as e:                                                                # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=work, args=(r,)) for r in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
        xy = np.stack([p[0] for p in parts]); inf = np.array([1 if p[1] else 0 for p in parts], dtype=np.uint8)
        return self.engines[0].points_sum(self.curve, xy, inf)

    def close(self):
        for eng, sh in zip(self.engines, self.shards):
            eng.free_shard(sh)


# ------------------------------------------------------------------------------------------------ coset-sharded d8 extension
def coset_shard_ids(world: int, rank: int, blowup: int = 8):
    """Cosets w_{blowup n}^r <w_n> evaluated by this rank (round-robin: one per GPU at world = blowup)."""
    return [r for r in range(blowup) if r % world == rank]


def interleave_cosets(per_rank, world: int, n: int, cols: int, blowup: int = 8):
    """per_rank[rk] = array [len(coset_shard_ids(world, rk)), cols, n, 4] (what the all-gather delivers) -> [cols, blowup * n, 4]
    with lde[c][blowup * i + r] = coset_r[c][i]."""
    out = np.zeros((cols, blowup * n, 4), dtype=np.uint64)
    for rk in range(world):
        for k, r in enumerate(coset_shard_ids(world, rk, blowup)):
            out[:, r::blowup, :] = per_rank[rk][k]
    return out
