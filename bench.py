#!/usr/bin/env python
"""bench.py -- depth frames/sec of the SimpleRecon hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A "step" is one pass of the hot path over one batch of synthetic keyframes that are already
resident in HBM.  One process per GPU (launched by torch.distributed.run for N > 1); keyframes
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="default: the BASELINE.json metric configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
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
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
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
