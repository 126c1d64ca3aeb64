"""The N > 1 job on the one GPU of the test box (VERDICT r02 "make N>1 real"): a plain
`python bench.py --gpus 2 --workload hero_cfg4_stream` -- no torchrun, the script launches its two ranks itself --
with SR_BENCH_SHARED_GPU=1 (both ranks on cuda:0, rendezvous + result gather over gloo because RCCL refuses two ranks
on one device).  The stream of 32 keyframes is sharded round-robin (keyframe i -> rank i mod 2), every depth map is
gathered to rank 0; the gathered [32,1,240,320] tensor must equal, bit for bit and in keyframe order, what ONE rank
computes for the same stream.  SR_GEMM_AUTOTUNE=0 in all processes: the image-prior encoder's library GEMMs otherwise
pick their algorithm by timing, per process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, steps, dump, shared):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SR_GEMM_AUTOTUNE="0", SR_BENCH_DUMP=dump, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if shared:
        env["SR_BENCH_SHARED_GPU"] = "1"
    else:
        env.pop("SR_BENCH_SHARED_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps),
                        "--warmup", "1", "--workload", "hero_cfg4_stream", "--no-roofline", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]       # rank 0 prints the one line
    return lines[0]


def test_two_ranks_on_one_gpu_equal_one_rank_bit_for_bit(tmp_path):
    two, one = str(tmp_path / "two.npy"), str(tmp_path / "one.npy")
    line2 = _run(2, 2, two, shared=True)
    assert line2["n_gpus"] == 2 and line2["steps"] == 2 and line2["config"]["frames_per_step_per_gpu"] == 8
    assert "not a measurement" in line2["config"]["parallelism"]
    line1 = _run(1, 4, one, shared=False)
    assert line1["n_gpus"] == 1
    a, b = np.load(two), np.load(one)
    assert a.shape == b.shape == (32, 1, 240, 320) and a.dtype == np.float32
    assert np.isfinite(a).all() and (a > 0).all()
    # keyframe order: frame i of the gathered tensor is keyframe i (rank i mod 2, position i // 2 of its shard)
    bad = [i for i in range(32) if not np.array_equal(a[i], b[i])]
    assert not bad, f"gathered keyframes differ from the single-rank stream: {bad}"
    assert not np.array_equal(a[0], a[1]) and not np.array_equal(a[0], a[2])   # distinct keyframes, not copies
