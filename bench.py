#!/usr/bin/env python
"""bench.py -- depth frames/sec of the SimpleRecon hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic keyframes that are already
resident in HBM.  One process per GPU: under torch.distributed.run the process is one rank; a plain
`python bench.py --gpus N` (no WORLD_SIZE in the environment) launches the N ranks itself and fails
loudly when fewer than N GPUs are visible -- it never reports an N-GPU line from one device.  Keyframes
are independent, so ranks shard them with no data-path collective ("scaling": "weak"); the only
exchange is the final gather of the depth maps to rank 0 over RCCL (inside the timed region).
Rank 0 prints ONE JSON line; see DESIGN.md for the roofline / cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3  # v_mfma_f32_32x32x2_f32 dense peak


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_command(n, argv, port=None):
    """The one-node launch of N ranks (one process per GPU) the driver would use: torch.distributed.run, rendezvous on
    127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
            os.path.abspath(__file__)] + list(argv)


def _shared_gpu():
    """SR_BENCH_SHARED_GPU=1 (tests only): the N ranks share the visible GPU(s) round-robin and rendezvous over gloo
    (RCCL refuses two ranks on one device) -- exercises the N > 1 job on a 1-GPU box.  Never a measurement."""
    return os.environ.get("SR_BENCH_SHARED_GPU", "0") == "1"


def _require_gpus(n):
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have == 0:
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if have < n and not _shared_gpu():
        raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible (one rank per GPU; refusing to report a "
                         f"{n}-GPU line from fewer devices)")
    return have


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="default: the BASELINE.json metric configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one per GPU) and return their exit code
        _require_gpus(args.gpus)
        import subprocess
        raise SystemExit(subprocess.call(launch_command(args.gpus, sys.argv[1:])))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    have = _require_gpus(world)
    shared = _shared_gpu() and world > 1
    local_dev = local_rank % have if shared else local_rank
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import bench_workloads
    name = args.workload or bench_workloads.DEFAULT
    wl = bench_workloads.WORKLOADS[name](dev, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for _ in range(args.warmup):
            wl.step()
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            wl.step(i)
        wl.finish(world)  # result gather to rank 0 (RCCL) -- part of the job
        barrier()
        elapsed = time.perf_counter() - t0
    dump = os.environ.get("SR_BENCH_DUMP")   # tests: the gathered result of the job, as rank 0 holds it
    if dump and rank == 0 and getattr(wl, "gathered", None) is not None:
        import numpy as np
        np.save(dump, wl.gathered.cpu().numpy())
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    frames = world * wl.frames_per_step * args.steps
    out = {
        "metric": "depth frames/sec (640x480, 7 src views, 64 planes)",
        "value": frames / elapsed,
        "unit": "frames/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": wl.config(world),
    }
    if shared:
        out["config"]["parallelism"] += " -- SR_BENCH_SHARED_GPU test mode: ranks share a device over gloo, not a measurement"
    if rank == 0:
        with torch.inference_mode():
            if not args.no_roofline:
                out["roofline"] = wl.roofline(args.steps)
                if hasattr(wl, "roofline_hbm"):
                    out["roofline_hbm"] = wl.roofline_hbm(args.steps)
                extra = wl.extra_kernels(args.steps)
                if extra:
                    out["kernels"] = extra
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = wl.cpu_baseline()
            except Exception as e:  # the baseline is reported next to the measurement, it must never cost the line
                out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                                       "sample": f"failed: {type(e).__name__}: {e}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
