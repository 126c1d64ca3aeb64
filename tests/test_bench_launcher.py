"""bench.py's N > 1 entry point: a plain `python bench.py --gpus N` must either launch N ranks or
fail loudly -- it must never print a line for fewer GPUs than asked."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SR_BENCH_SHARED_GPU")}
    return env


def test_launch_command_is_one_rank_per_gpu_on_loopback():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "5"], port=29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5"]


def test_gpus_2_without_torchrun_never_reports_one_gpu():
    """On a box with fewer than 2 GPUs (this container: none) the command exits non-zero with the reason and prints
    no JSON line at all; with >= 2 GPUs it would have become the launcher of 2 ranks."""
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a box with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU" in (r.stderr + r.stdout)
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            assert json.loads(line).get("n_gpus") != 1, "bench.py --gpus 2 reported a 1-GPU line"


def test_world_size_mismatch_is_an_error():
    env = dict(_clean_env(), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_rank_pinning_gives_local_ranks_disjoint_core_slices():
    """bench.pin_rank: the local ranks of a node take disjoint, equal slices of the allowed cores (no NUMA information
    without a GPU: the slice is cut from everything the process may use) and torch's intra-op pool shrinks with it."""
    sys.path.insert(0, ROOT)
    code = ("import sys, os, json; sys.path.insert(0, %r); import bench; "
            "r = bench.pin_rank(int(sys.argv[1]), 4, 0); r['set'] = sorted(os.sched_getaffinity(0)); print(json.dumps(r))" % ROOT)
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        import pytest
        pytest.skip("needs >= 4 cores")
    sets = []
    for rank in range(4):
        r = subprocess.run([sys.executable, "-c", code, str(rank)], env=_clean_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert d["pinned"] and d["cores"] == len(allowed) // 4 == len(d["set"]) and 1 <= d["torch_threads"] <= 16
        sets.append(set(d["set"]))
    assert all(sets[i].isdisjoint(sets[j]) for i in range(4) for j in range(i))
    assert bench_cpu_list() == [0, 1, 2, 3, 8, 10, 11]


def bench_cpu_list():
    import bench
    return bench._cpu_list("0-3,8,10-11")


def test_fenced_experiment_field_never_costs_the_line(monkeypatch):
    """bench._fenced_experiment parses the child's JSON line into a `"headline": false` record and turns any failure of the
    child (crash, time-out, no line) into a record that says so."""
    import json
    import subprocess

    import bench

    class R:
        stdout = 'noise\n' + json.dumps({"metric": "m", "value": 400.0, "unit": "frames/s", "ms_per_step": 20.0, "steps": 10,
                                         "dtype": "f32 I/O ...; 2 x f16 pieces"}).replace('{"metric"', '{"metric"') + "\n"
    monkeypatch.setattr(subprocess, "run", lambda *a, **k: R())
    rec = bench._fenced_experiment("hero_cfg3_f16x3_convs", 10, 3)
    assert rec["headline"] is False and rec["value"] == 400.0 and rec["ms_per_step"] == 20.0 and "f16" in rec["dtype"]

    def boom(*a, **k):
        raise subprocess.TimeoutExpired(cmd="bench", timeout=180)
    monkeypatch.setattr(subprocess, "run", boom)
    rec = bench._fenced_experiment("hero_cfg3_f16x3_convs", 10, 3)
    assert rec["headline"] is False and rec["value"] is None and "TimeoutExpired" in rec["note"]
