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

# The step runs on 4 HIP streams (main, image-prior encoder, two decoder branches); an RCCL process group brings its own.  With
# ROCm's default of 4 hardware queues per process the fifth stream shares a queue with one of ours and the streams serialise
# against each other: +0.45 ms per step (1.6 %) from nothing but `init_process_group("nccl")` (scripts/pg_probe.py, r04).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def _cpu_list(spec):
    """'0-15,128-143' -> [0..15, 128..143]"""
    out = []
    for part in spec.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def _gpu_numa_cpus(local_dev):
    """The host cores of the NUMA node GPU `local_dev` hangs off: PCI bus id (torch) -> /sys/bus/pci/devices/<id>/
    numa_node -> /sys/devices/system/node/node<N>/cpulist.  None when sysfs does not say (containers, one-node hosts)."""
    try:
        bus = torch.cuda.get_device_properties(local_dev).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(local_dev), "pci_domain_id", 0)
        devid = getattr(torch.cuda.get_device_properties(local_dev), "pci_device_id", 0)
        path = f"/sys/bus/pci/devices/{dom:04x}:{bus:02x}:{devid:02x}.0/numa_node"
        with open(path) as f:
            node = int(f.read().strip())
        if node < 0:
            return None
        with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
            return _cpu_list(f.read())
    except Exception:
        return None


def pin_rank(local_rank, local_world, local_dev, threads=None):
    """One eager Python rank per GPU issues ~800 launches per step: keep each rank's launch thread on its GPU's NUMA
    node and away from the other ranks' cores.  The rank takes the `local_rank`-th of `local_world` equal slices of the
    cores it may use (its GPU's NUMA node when sysfs knows it and shares cores with the allowed set, else everything
    the process is allowed); torch's intra-op pool is cut to the slice too (SR_BENCH_PIN=0 turns all of it off).
    Returns what it did, for the bench line."""
    if os.environ.get("SR_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False}
    allowed = sorted(os.sched_getaffinity(0))
    numa = _gpu_numa_cpus(local_dev)
    pool = sorted(set(allowed) & set(numa)) if numa else []
    by_numa = bool(pool)
    if by_numa:
        # ranks on the same NUMA node split that node's cores
        peers = [r for r in range(local_world) if (_gpu_numa_cpus(r) or []) == numa] if local_world > 1 else [local_rank]
        idx, n = (peers.index(local_rank), len(peers)) if local_rank in peers else (0, 1)
    else:
        pool, idx, n = allowed, local_rank, max(local_world, 1)
    per = max(1, len(pool) // n)
    mine = pool[idx * per:(idx + 1) * per] or pool
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {"pinned": False}
    nthreads = threads or min(len(mine), 16)
    torch.set_num_threads(max(1, nthreads))
    return {"pinned": True, "cores": len(mine), "first_core": mine[0], "by_numa_node": by_numa,
            "torch_threads": torch.get_num_threads()}


def _fenced_experiment(workload, steps, warmup):
    """One fenced workload timed in a CHILD process after the headline has been measured -- its own
    switches, its own model, and a crash or a time-out there costs this field, never the line.  Reported beside the headline
    with the arithmetic it ran on; it is not the headline and `value` / `ms_per_step` of the line do not see it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", workload, "--steps", str(steps), "--warmup",
                            str(warmup), "--no-cpu-baseline", "--no-roofline", "--no-experiments"], capture_output=True,
                           text=True, timeout=180, env={k: v for k, v in os.environ.items()
                                                        if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")})
        line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][-1]
        d = json.loads(line)
        return {"workload": workload, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                "dtype": d["dtype"], "headline": False,
                "note": "fenced experiment, off by default: same model, inputs and parity tests as the headline; the MLP sweep's layers "
                        "1-2 and the Winograd convolutions multiply two fp16 pieces per fp32 operand (three exact products)"}
    except Exception as e:   # never costs the line
        return {"workload": workload, "value": None, "headline": False, "note": f"not measured: {type(e).__name__}: {e}"[:300]}


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
    ap.add_argument("--no-experiments", action="store_true", help="(default since r05; kept for old command lines)")
    ap.add_argument("--experiments", action="store_true",
                    help="also report the fenced split-precision workload in an `experiments` field (a child process after the "
                         "headline is timed; narrower arithmetic than the reference's fp32: never the headline -- VERDICT r04)")
    ap.add_argument("--force-collective", action="store_true",
                    help="initialise the process group and run the result gather / all-reduce / barrier collectives even "
                         "with one rank (exercises the RCCL path on a 1-GPU box)")
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
    collective = world > 1 or args.force_collective
    # RCCL prints a version banner on STDOUT when its first communicator comes up ("RCCL version : ..."): the contract is ONE JSON
    # line there.  Everything the process (and the libraries under it) writes to file descriptor 1 goes to stderr instead; the
    # JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    pin = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local_dev) if collective else {"pinned": False}

    import bench_workloads
    name = args.workload or bench_workloads.DEFAULT
    wl = bench_workloads.WORKLOADS[name](dev, rank)

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.inference_mode():
        for _ in range(args.warmup):
            wl.step()
        if collective and args.warmup > 0:
            # the job's gather once, untimed: RCCL sets its point-to-point channels up at the first use of a collective
            # (~10 ms), which is start-up cost, not part of a steady-state step; the timed region below gathers again
            wl.finish(world, force_collective=args.force_collective)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            wl.step(i)
        t_loop = time.perf_counter() - t0
        wl.finish(world, force_collective=args.force_collective)  # result gather to rank 0 (RCCL) -- part of the job
        t_fin = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
        if os.environ.get("SR_BENCH_TRACE"):
            print(f"[rank {rank}] steps issued {t_loop*1e3:.2f} ms, + gather issued {t_fin*1e3:.2f} ms, + barrier/sync {elapsed*1e3:.2f} ms",
                  file=sys.stderr, flush=True)
    dump = os.environ.get("SR_BENCH_DUMP")   # tests: the gathered result of the job, as rank 0 holds it
    if dump and rank == 0 and getattr(wl, "gathered", None) is not None:
        import numpy as np
        np.save(dump, wl.gathered.cpu().numpy())
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if shared else dev)
    if collective:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    _flush_c_stdio()   # (every rank, before rank 0 prints: whatever the collective library buffered goes out ahead of the JSON line)

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
        "dtype": getattr(wl, "dtype", "f32"),
        "data": "synthetic",
        "config": wl.config(world),
    }
    if collective:
        out["config"]["backend"] = dist.get_backend()
        out["config"]["rank_pinning"] = pin
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
        if world == 1 and name == bench_workloads.DEFAULT and args.experiments and not args.no_experiments:
            out["experiments"] = [_fenced_experiment("hero_cfg3_f16x3_convs", min(args.steps, 10), args.warmup)]
        _flush_c_stdio()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if collective:
        dist.barrier()
        dist.destroy_process_group()


def _flush_c_stdio():
    """RCCL writes its version banner through C stdio; on a pipe that buffer is flushed at process exit, i.e. AFTER the JSON
    line.  Flushing it first keeps the JSON line the last line of rank 0's stdout."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


if __name__ == "__main__":
    main()
