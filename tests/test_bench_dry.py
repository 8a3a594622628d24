"""bench.py's launcher and N > 1 plumbing in the CPU test tier: `python bench.py --gpus 2 --dry-run` re-executes itself under
the torch.distributed launcher (bench.py: os.execv ... torch.distributed.run), the two ranks form a gloo process group, take their point ranges, run the
timed loop's submit / wait / combine logic three times with a stand-in shard (every MSM "returns" the blinding base h; no GPU work), all-gather and fold
the partial sums with the product's kh_points_sum, take the maximum over ranks of the region time and rank 0 prints ONE line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(gpus, extra_env=None):
    env = dict(os.environ, KH_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "5", "--warmup", "1", "--dry-run"],
                          env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)


def _run(gpus, extra_env=None):
    r = _launch(gpus, extra_env)
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


def test_bench_dry_two_ranks_carry_both_scalings_on_one_line():
    """Round 6 (d): at N > 1 one invocation measures the weak workload AND BASELINE config 4 (one 2^22-point MSM cut over the ranks); the latter rides in the
    line as `strong`, with its own regions, and its combined result really went through the collective and the fold."""
    line, _ = _run(2)
    st = line["strong"]
    assert st["scaling"] == "strong" and st["points_total"] == 1 << 22 and st["points_per_gpu"] == 1 << 21 and len(st["value_runs"]) == 3
    assert st["combined_result_is_world_times_h"] is True and st["collective_backend"] == "gloo-torch"
    line, _ = _run(2, {"KH_BENCH_NO_STRONG": "1"})
    assert line["strong"] is None


def test_bench_dry_combine_runs_off_the_submit_thread():
    """Round 6 (a): the collective + fold of finished MSMs run on a combiner thread in submission order (the submitting thread never sits in a collective);
    KH_BENCH_INLINE_COMBINE=1 restores the inline combine -- same combined result either way."""
    line, _ = _run(2)
    assert line["combine_thread"] is True and line["combined_result_is_world_times_h"] is True
    line, _ = _run(2, {"KH_BENCH_INLINE_COMBINE": "1"})
    assert line["combine_thread"] is False and line["combined_result_is_world_times_h"] is True
    line, _ = _run(2, {"KH_BENCH_COMBINE_EVERY": "2"})              # two partial sums per collective
    assert line["combined_result_is_world_times_h"] is True and line["config"]["partials_per_collective"] == 2


def test_bench_dry_failed_rank_ends_the_run_with_an_error_line():
    """Round 6 (c): a rank that raises ends the whole run -- a JSON line with `error` on stdout, no hang (the launcher takes the other rank down)."""
    r = _launch(2, {"KH_BENCH_FAIL_RANK": "1"})
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert lines and all("error" in l and l["value"] is None for l in lines)
    assert any(l["error_rank"] == 1 and "fails on purpose" in l["error"] for l in lines)
    assert r.returncode != 0


def test_bench_dry_watchdog_turns_a_hang_into_an_error_line():
    """Round 6 (c): every phase of bench.py runs under a deadline; one that passes prints the phase's name in a JSON `error` line and ends the rank."""
    r = _launch(2, {"KH_BENCH_WATCHDOG_S": "0.0000001"})
    lines = [json.loads(l) for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert lines and all("watchdog" in l["error"] for l in lines)
    assert r.returncode != 0


def test_gpu_locality_parser(tmp_path, monkeypatch):
    """Round 6 (b): NUMA node and CPU list of a GPU from sysfs (bench.py binds a rank's host threads to them at N > 1 and prints them in the rank banner)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert bench.gpu_locality("ffff:ff:1f.0") == (None, None)           # no such device: nothing is bound
    import builtins
    real_open = builtins.open
    fake = {"/sys/bus/pci/devices/0000:c1:00.0/numa_node": "1\n", "/sys/bus/pci/devices/0000:c1:00.0/local_cpulist": "64-67,192-193\n"}

    def fopen(path, *a, **k):
        if path in fake:
            import io
            return io.StringIO(fake[path])
        return real_open(path, *a, **k)
    monkeypatch.setattr(builtins, "open", fopen)
    assert bench.gpu_locality("0000:C1:00.0") == (1, {64, 65, 66, 67, 192, 193})
