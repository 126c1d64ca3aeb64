"""Workloads of bench.py (kept outside the product package because the cpu_baseline leg
imports the oracle, which only tests / smoke / bench may do)."""
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from simplerecon_amd import _lib, synthetic  # noqa: E402
from simplerecon_amd.cost_volume import CostVolumeManager  # noqa: E402

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TF = 157.3


def _time_launches(fn, n):
    """Average device time of fn() over n launches, HIP events on the launch stream."""
    fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(n):
        fn()
    end.record()
    end.synchronize()
    return start.elapsed_time(end) * 1e-3 / n


def _pmc_traffic(tag):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/traffic.json), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path)).get(tag)
        except Exception:
            return None
    return None


class DotCfg2:
    """BASELINE.json configs[1]: dot_product_model, batch 1, 7 source views, 64 planes, 640x480
    (matching resolution 120x160, 16 channels) -- the fused warp + dot + view-reduce kernel."""
    name = "dot_cfg2"
    B, K, Cc, D, h, w = 1, 7, 16, 64, 120, 160

    def __init__(self, dev, rank, B=None):
        if B is not None:
            self.B = B
        self.dev = dev
        self.frames_per_step = self.B
        self.inp = synthetic.cost_volume_inputs(self.B, self.K, self.Cc, self.h, self.w, seed=rank, device=dev)
        self.mgr = CostVolumeManager(self.h, self.w, num_depth_bins=self.D).to(dev)
        self.last = None

    def step(self, i=0):
        self.last = self.mgr(**self.inp)

    def finish(self, world):
        if world > 1:
            lowest = self.last[1].contiguous()
            out = [torch.empty_like(lowest) for _ in range(world)] if dist.get_rank() == 0 else None
            dist.gather(lowest, out, dst=0)

    def config(self, world):
        return {"workload": f"{self.name}: CostVolumeManager (dot-product plane sweep), batch {self.B}/GPU, "
                            f"{self.K} source views, {self.D} planes, 640x480 image -> {self.h}x{self.w} matching "
                            f"features x {self.Cc} ch, fp32; cost volume only (BASELINE.json configs[1])",
                "frames_per_step_per_gpu": self.B, "parallelism": f"replica x{world} (keyframes sharded)"}

    def algorithmic_bytes(self):
        N = self.h * self.w
        per_frame = 4 * ((self.K + 1) * self.Cc * N + self.D * N + N) + 4 * (32 * self.K + 16 + self.D)
        return per_frame * self.B

    def roofline(self, n):
        lib = _lib.lib()
        m, inp = self.mgr, self.inp
        B, K, Cc, h, w, D = self.B, self.K, self.Cc, self.h, self.w, self.D
        planes = m.generate_depth_planes(B, inp["min_depth"], inp["max_depth"])
        vol = torch.empty((B, D, h, w), device=self.dev)
        lowest = torch.empty((B, h, w), device=self.dev)
        ws = torch.empty(lib.sr_volume_workspace_bytes(B, K, Cc, h, w), dtype=torch.uint8, device=self.dev)
        st = _lib.stream_ptr(self.dev)
        rc = lib.sr_volume_prepare(_lib.ptr(inp["src_feats"]), _lib.ptr(inp["src_Ks"]), _lib.ptr(inp["src_extrinsics"]),
                                   None, B, K, Cc, h, w, _lib.ptr(ws), ws.numel(), st)
        _lib.check(rc, "sr_volume_prepare")

        def sweep():
            rc = lib.sr_dot_volume_sweep(_lib.ptr(inp["cur_feats"]), _lib.ptr(inp["cur_invK"]), _lib.ptr(planes),
                                         *planes.stride(), B, K, Cc, h, w, D, _lib.ptr(vol), D * h * w, h * w, 1,
                                         _lib.ptr(lowest), None, _lib.ptr(ws), ws.numel(), st)
            _lib.check(rc, "sr_dot_volume_sweep")
        t = _time_launches(sweep, max(n, 20))
        nbytes = self.algorithmic_bytes()
        achieved = nbytes / t / 1e9
        N = h * w
        return {"kernel": "sr_dot_volume_kernel<16>", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": _pmc_traffic(self.name),
                "avg_launch_us": t * 1e6, "algorithmic_bytes_per_launch": nbytes,
                "onchip_gather_GBps": B * D * K * N * 4 * Cc * 4 / t / 1e9}

    def extra_kernels(self, n):
        return None

    def cpu_baseline(self):
        import oracle
        n = {k: v.cpu().numpy() for k, v in self.inp.items()}
        planes = self.mgr.generate_depth_planes(1, self.inp["min_depth"], self.inp["max_depth"])[:, :, 0, 0].cpu().numpy()
        one = {k: (v[:1] if v.ndim > 0 and v.shape[0] == self.B and k not in ("min_depth", "max_depth") else v)
               for k, v in n.items()}

        def run():
            oracle.dot_volume(one["cur_feats"], one["src_feats"], one["src_Ks"], one["src_extrinsics"],
                              one["cur_invK"], planes)
        run()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 or reps < 3:
            run()
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {"value": 1.0 / dt, "unit": "frames/s", "cores": oracle.num_threads(), "kind": "port",
                "sample": f"{reps} repetitions of 1 frame of {self.name} through oracle/sr_oracle_dot_volume_f32 "
                          f"(plain C + OpenMP, {oracle.num_threads()} threads of {os.cpu_count()} host CPUs)"}


WORKLOADS = {
    "dot_cfg2": lambda dev, rank: DotCfg2(dev, rank),
    "dot_b8": lambda dev, rank: DotCfg2(dev, rank, B=8),
}
DEFAULT = "dot_cfg2"
