"""First contact with RCCL on the one GPU of the test box.  Every other multi-rank test
runs gloo (RCCL refuses two ranks on one device) and a world of one used to return from `sharding.gather_results`
before any collective ran -- so the "nccl" branches (`init_process_group("nccl", device_id=...)`, the device-buffer
`dist.gather`, `all_reduce(MAX)` on a device scalar, `barrier`) would have executed for the first time on the driver's
8-GPU box.  Here they run with world = 1 on cuda:0, in child processes (a process group is process-global state)."""
import json
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    return env


def test_gather_results_over_nccl_world_of_one():
    code = textwrap.dedent(f"""
        import os, sys, socket
        sys.path.insert(0, {ROOT!r})
        import torch, torch.distributed as dist
        from simplerecon_amd import sharding
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=dev)
        assert dist.get_backend() == "nccl"
        g = torch.Generator(device="cpu").manual_seed(3)
        local = torch.randn((5, 1, 24, 32), generator=g).to(dev)
        # the early return: no collective, same tensor object
        assert sharding.gather_results(local, 5) is local
        out = sharding.gather_results(local, 5, dst=0, force_collective=True)
        assert out is not local and out.is_cuda and out.shape == local.shape
        torch.cuda.synchronize()
        assert torch.equal(out, local)
        # bench.py's other collectives: MAX all-reduce of the elapsed time on a device scalar, barrier
        t = torch.tensor([1.25], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.25
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        print("RCCL_OK")
    """)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _bench(extra, dump):
    env = _env()
    env["SR_BENCH_DUMP"] = dump
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                        "--workload", "hero_cfg4_stream", "--no-roofline", "--no-cpu-baseline"] + extra,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return lines[0]


def test_bench_job_through_rccl_equals_the_plain_one_rank_job(tmp_path):
    """`python bench.py --gpus 1 --force-collective`: process group on "nccl", rank pinning, barrier on both sides of
    the timed region, the gather of every depth map of the stream through dist.gather on device buffers, MAX
    all-reduce of the time -- and the gathered stream is, bit for bit, what the collective-free job produces."""
    a_path, b_path = str(tmp_path / "coll.npy"), str(tmp_path / "plain.npy")
    a = _bench(["--force-collective"], a_path)
    assert a["config"]["backend"] == "nccl" and a["n_gpus"] == 1
    assert "rank_pinning" in a["config"]
    b = _bench([], b_path)
    assert "backend" not in b["config"]
    x, y = np.load(a_path), np.load(b_path)
    assert x.shape == y.shape == (16, 1, 240, 320)
    assert np.array_equal(x, y)
