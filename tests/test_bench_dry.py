"""bench.py's launcher and N > 1 plumbing in the CPU test tier: `python bench.py --gpus 2 --dry-run` re-executes itself under
the torch.distributed launcher (bench.py: os.execv ... torch.distributed.run), the two ranks form a gloo process group, take their point ranges, run the
timed loop's submit / wait / combine logic three times with a stand-in shard (every MSM "returns" the blinding base h; no GPU work), all-gather and fold
the partial sums with the product's kh_points_sum, take the maximum over ranks of the region time and rank 0 prints ONE line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(gpus):
    env = dict(os.environ, KH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "5", "--warmup", "1", "--dry-run"],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()
    return json.loads(lines[0]), r.stderr.decode()


def test_bench_dry_two_ranks():
    line, err = _run(2)
    assert line["dry_run"] and line["n_gpus"] == 2 and line["ranks_reported"] == 2 and line["steps"] == 5
    assert line["combined_result_is_world_times_h"] is True
    assert line["config"]["collective_backend"] == "gloo-torch" and line["config"]["world_size_seen"] == 2 and line["config"]["partials_per_collective"] == 1
    assert len(line["value_runs"]) == 3
    assert "[bench rank 0/2]" in err and "[bench rank 1/2]" in err                # the per-rank diagnostics a first multi-GPU run is debugged from


def test_bench_dry_one_rank():
    line, _ = _run(1)
    assert line["n_gpus"] == 1 and line["combined_result_is_world_times_h"] is True
