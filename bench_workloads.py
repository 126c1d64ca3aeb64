"""Workloads of bench.py (kept outside the product package because the cpu_baseline leg
imports the oracle, which only tests / smoke / bench may do)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

from simplerecon_amd import _lib, sharding, synthetic  # noqa: E402
from simplerecon_amd.cost_volume import CostVolumeManager  # noqa: E402

HBM_PEAK_GBS = 8000.0
FP32_MFMA_PEAK_TF = 157.3
F16_MFMA_PEAK_TF = 2500.0   # dense bf16 / f16 MFMA (MI355X_MICROARCH.md)


def _mlp_kernel_name(K):
    """Which sr_mlp_volume_kernel<W1_LDS, W2_LDS> sr_mlp_volume_sweep dispatches (mirrors sr_mlp_volume.hip)."""
    split = _lib.split_mode_name("SR_MLP_SPLIT")
    if split in ("bf16", "f16"):
        return f"sr_mlp_volume_split_kernel<{1 if split == 'bf16' else 2}>"
    w1, w2, w3 = 12 * K * 1024, 65 * 1024, 1024
    if w1 + w2 + w3 <= 160 * 1024:
        return "sr_mlp_volume_kernel<true, true>"
    if w1 + w3 <= 160 * 1024:
        return "sr_mlp_volume_kernel<true, false>"
    return "sr_mlp_volume_kernel<false, true>"


def _best_cpu_threads(K=7, C=16, h=120, w=160, table=None):
    """Thread count for the ATen CPU baseline, probed ON THE WORKLOAD'S OWN PLANE SIZE.  ATen's intra-op scaling on a many-core
    host is far from monotonic and depends on the tensor size (r04: a 60x80 probe picked 64 threads on the driver's box where
    16 is 2x faster on the real 120x160 sweep), so every count of (4, 8, 16, 32, 64, 128) that the host has is timed on two
    planes of the real sweep -- best of two runs each, no early exit -- and the fastest wins.  `table` (a dict) receives the
    per-count seconds per plane, which the bench line reports."""
    import bench_cpu_aten as aten
    inp = synthetic.cost_volume_inputs(1, K, C, h, w, seed=0)
    cin = C * (K + 1) + 10 * K + 4
    lin = [(torch.randn(128, cin), torch.zeros(128)), (torch.randn(128, 128), torch.zeros(128)),
           (torch.randn(1, 128), torch.zeros(1))]
    planes = torch.tensor([[1.0, 2.0]])
    best, best_t = None, None
    ncpu = os.cpu_count() or 1
    for nt in (4, 8, 16, 32, 64, 128):
        if nt > ncpu and best is not None:
            break
        torch.set_num_threads(min(nt, ncpu))
        with torch.inference_mode():
            args = (inp["cur_feats"], inp["src_feats"], inp["src_Ks"], inp["src_extrinsics"], inp["src_poses"],
                    inp["cur_invK"], planes, lin)
            aten.mlp_volume(*args)   # warm-up: thread pool, oneDNN primitives
            ts = []
            for _ in range(2):
                t0 = time.perf_counter()
                aten.mlp_volume(*args)
                ts.append((time.perf_counter() - t0) / planes.shape[1])
        t = min(ts)
        if table is not None:
            table[str(min(nt, ncpu))] = round(t, 4)
        if best_t is None or t < best_t:
            best, best_t = min(nt, ncpu), t
    torch.set_num_threads(best or 1)
    return best or 1


def _dot_kernel_name(B, h, w, D):
    """Which sr_dot_volume_lds_kernel<CAP, WPS, G> sr_dot_volume_sweep dispatches for C = 16 (mirrors
    sr_launch_dot_volume_lds in csrc/sr_dot_volume_lds.hip); the L1-gather kernel when SR_DOT_LDS=0."""
    if _lib.get_option("SR_DOT_LDS") == 0:
        return "sr_dot_volume_kernel16q"
    cap = 770 if _lib.get_option("SR_DOT_LDS_CAP") == 770 else 634
    g = _lib.get_option("SR_DOT_LDS_G")
    if g not in (2, 4, 8):
        g = 2 if B * ((w + 31) // 32) * ((h + 7) // 8) * ((D + 3) // 4) < 2048 else 4
    return f"sr_dot_volume_lds_kernel<{cap}, {3 if cap == 770 else 4}, {g}>"


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


def _pmc_algorithmic_bytes(tag):
    """Algorithmic bytes per launch recorded next to the PMC traffic of the same kernel (profiles/traffic.json)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        v = json.load(open(path)).get(tag)
        return v.get("algorithmic_bytes") if isinstance(v, dict) else None
    except Exception:
        return None


def _pmc_traffic(tag, kernel=None):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/traffic.json), or None -- also None when
    the entry was measured on another kernel than the one this run launches (an ablation such as SR_DOT_LDS=0)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            v = json.load(open(path)).get(tag)
            if isinstance(v, dict):
                if kernel and v.get("kernel") and not (kernel.startswith(v["kernel"].rstrip(">")) or
                                                       v["kernel"].startswith(kernel.rstrip(">"))):
                    return None
                return v.get("bytes")
            return v
        except Exception:
            return None
    return None


class DotCfg2:
    """BASELINE.json configs[1]: dot_product_model, batch 1, 7 source views, 64 planes, 640x480
    (matching resolution 120x160, 16 channels) -- the fused warp + dot + view-reduce kernel."""
    name = "dot_cfg2"
    B, K, Cc, D, h, w = 1, 7, 16, 64, 120, 160

    def __init__(self, dev, rank, B=None, D=None, name=None):
        if B is not None:
            self.B = B
        if D is not None:
            self.D = D
        if name is not None:
            self.name = name
        self.dev = dev
        self.frames_per_step = self.B
        self.inp = synthetic.cost_volume_inputs(self.B, self.K, self.Cc, self.h, self.w, seed=rank, device=dev)
        self.mgr = CostVolumeManager(self.h, self.w, num_depth_bins=self.D).to(dev)
        self.last = None

    def step(self, i=0):
        self.last = self.mgr(**self.inp)

    def finish(self, world, force_collective=False):
        # the job's only exchange: this step's results to rank 0 (keyframe i lives on rank i mod world)
        sharding.gather_results(self.last[1], world * self.last[1].shape[0], dst=0, force_collective=force_collective)

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
        # the launch = [memset of the argmax keys] + sweep kernel + [key -> depth kernel]; the sweep kernel alone is timed
        # by rocprofv3 (profiles/): subtract nothing here, report both
        nbytes = self.algorithmic_bytes()
        achieved = nbytes / t / 1e9
        N = h * w
        traffic = _pmc_traffic(self.name, _dot_kernel_name(B, h, w, D))
        # The bound that actually applies (from the r02 PMC instruction counts and the VALU issue-rate
        # microbenchmark): per frame at 64 planes x 7 views the instruction stream needs ~10 us of issue slots (16 resident
        # waves per CU x one instruction per ~5 cycles per wave) and the LDS tap reads ~8 us (1.1 GB after culling at
        # <= 256 B/clk/CU); both scale with the (pixel, plane, view) sample count.
        samples = float(B) * D * K * N
        ref_samples = 64.0 * 7 * 19200
        issue_us, lds_us = 10.0 * samples / ref_samples, 8.0 * samples / ref_samples
        return {"kernel": _dot_kernel_name(B, h, w, D), "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_over_algorithmic": (traffic / nbytes) if traffic else None,
                "avg_launch_us": t * 1e6, "algorithmic_bytes_per_launch": nbytes,
                "issue_lds_bound": {"issue_us": issue_us, "lds_us": lds_us, "floor_us": issue_us + lds_us,
                                    "frac": (issue_us + lds_us) / (t * 1e6),
                                    "note": "instruction-issue + LDS-tap floor of the bit-exact fp32 sweep (DESIGN.md 3.1) "
                                            "over the measured launch: the fraction of the bound that applies; the HBM "
                                            "figures above are the north star's metric on compulsory bytes"},
                "lds_tap_GBps": B * D * K * N * 4 * Cc * 4 / t / 1e9,
                "note": "achieved = algorithmic (compulsory) bytes / time of one sr_dot_volume_sweep call (keys memset + "
                        "LDS-staged sweep kernel + key-to-depth kernel), HIP events on the launch stream; the sweep is "
                        "bound by instruction issue + LDS tap reads (4 taps x 64 B per (pixel, plane, view) sample = "
                        "lds_tap_GBps if no view were culled), not by HBM: see DESIGN.md 3.1"}

    def extra_kernels(self, n):
        return None

    def cpu_baseline(self):
        """The reference's CPU PyTorch path for this workload = the ATen operator sequence of CostVolumeManager
        (bench_cpu_aten.dot_volume) on all host cores, 1 frame, repeated for ~10 s."""
        import bench_cpu_aten as aten
        _best_cpu_threads()
        c = {k: (v[:1].cpu() if v.dim() > 0 and v.shape[0] == self.B and k not in ("min_depth", "max_depth") else v.cpu())
             for k, v in self.inp.items()}
        planes = self.mgr.generate_depth_planes(1, self.inp["min_depth"], self.inp["max_depth"])[:, :, 0, 0].cpu().contiguous()

        def run():
            with torch.inference_mode():
                aten.dot_volume(c["cur_feats"], c["src_feats"], c["src_Ks"], c["src_extrinsics"], c["cur_invK"], planes)
        run()
        reps, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 10.0 or reps < 3:
            run()
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return {"value": 1.0 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample": f"{reps} repetitions of 1 frame of {self.name} through bench_cpu_aten.dot_volume: the ATen "
                          f"(PyTorch {torch.__version__} CPU, fp32) operator sequence of the reference's "
                          f"CostVolumeManager (matmul projection, F.grid_sample, mul / sum per plane), "
                          f"torch.set_num_threads({torch.get_num_threads()}) = the fastest of 8..128 on this host ({os.cpu_count()} CPUs)"}


class HeroCfg3:
    """BASELINE.json configs[2]: hero_model.yaml (metadata-MLP matching), batch 8, 7 source views,
    64 planes, 640x480: the whole DepthModel.forward = image-prior encoder (EfficientNetV2-S pyramid) on the B
    reference images + ResnetMatchingEncoder on the B*(1+K) images -> FeatureVolumeManager sweep -> CVEncoder ->
    DepthDecoderPP -> exp, all on hand-written HIP kernels.  Inputs = the images and camera matrices, resident in HBM.
    prior=False ("*_noprior") feeds a synthetic image-prior pyramid instead of running that encoder (the r01 workload
    definition before the encoder was native); with_encoder=False ("*_core") also starts from synthetic matching
    features."""
    name = "hero_cfg3"
    B, K, Cc, D, h, w = 8, 7, 16, 64, 120, 160
    feature_volume_type = "mlp_feature_volume"

    dtype = "f32"
    ROTATE = 3   # resident input batches step(i) cycles through

    def __init__(self, dev, rank, B=None, streams=1, with_encoder=True, name=None, graph=False, prior=None, split=None, split_convs=False):
        from simplerecon_amd import depth_model as dm
        self.prior = with_encoder if prior is None else (prior and with_encoder)
        if split is not None:
            # FENCED EXPERIMENTS, never the headline: layers 1-2 of the metadata-MLP sweep -- and with
            # split_convs the Winograd 3x3 convolutions -- multiply on the 16-bit matrix pipe, every fp32 operand as two 16-bit
            # pieces, three products, fp32 accumulate.  The switches are read by the library per call; a bench process runs
            # one workload.
            _lib.set_option("SR_MLP_SPLIT", split)
            if split_convs:
                _lib.set_option("SR_WINO_SPLIT", split)
            self.dtype = (f"f32 I/O and accumulate; MLP-sweep layers 1-2{' and the Winograd 3x3 convolutions' if split_convs else ''}"
                          f": operands as 2 x {split} pieces, 3 MFMA products (fenced experiment, not the headline arithmetic)")
        if B is not None:
            self.B = B
        if name is not None:
            self.name = name
        self.with_encoder = with_encoder
        self.use_graph = graph
        self._graphed = None
        self.streams = streams
        self.dev = dev
        self.frames_per_step = self.B
        opts = dm.default_options(image_width=4 * self.w, image_height=4 * self.h, model_num_views=self.K + 1,
                                  matching_num_depth_bins=self.D, feature_volume_type=self.feature_volume_type)
        model = dm.DepthModel(opts)   # both encoders native; "*_noprior" / "*_core" simply do not run them
        if with_encoder:
            synthetic.seeded_fill_(model.matching_model, seed=4)
        if self.prior:
            synthetic.seeded_fill_(model.encoder, seed=5)
        synthetic.seeded_fill_(model.cost_volume_net, seed=1)
        synthetic.seeded_fill_(model.depth_decoder, seed=2)
        if hasattr(model.cost_volume, "mlp"):
            synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
        self.model = model.to(dev).eval()
        self.model.num_streams = streams
        if os.environ.get("SR_PRIOR_SIDE"):   # experiment switch: image-prior encoder on the main stream (0) / a side stream (1)
            self.model.prior_on_side_stream = os.environ["SR_PRIOR_SIDE"] != "0"
        inp = synthetic.cost_volume_inputs(self.B, self.K, self.Cc, self.h, self.w, seed=rank, device=dev)
        self.inp = inp
        self.pyramid = [f.contiguous(memory_format=torch.channels_last) for f in
                        synthetic.image_prior_pyramid(self.B, self.h, self.w, seed=rank, device=dev)]
        if with_encoder:  # ImageNet-normalised images ~ N(0,1)
            g = torch.Generator(device="cpu").manual_seed(1000 + rank)
            self.cur_image = torch.randn((self.B, 3, 4 * self.h, 4 * self.w), generator=g).to(dev)
            self.src_image = torch.randn((self.B, self.K, 3, 4 * self.h, 4 * self.w), generator=g).to(dev)
        # step(i) rotates over ROTATE resident batches (VERDICT r05 hygiene (a): the 236 MB of images of ONE batch are about the
        # size of the Infinity Cache, so 20 steps on the same batch could have read them from there).  Batch 0 is the one the
        # parity / profile helpers use; the others differ in images, poses and intrinsics.
        self.batches = [(self.inp, getattr(self, "cur_image", None), getattr(self, "src_image", None))]
        for j in range(1, self.ROTATE if with_encoder else 1):
            inp_j = synthetic.cost_volume_inputs(self.B, self.K, self.Cc, self.h, self.w, seed=rank + 17 * j, device=dev)
            self.batches.append((inp_j, torch.randn((self.B, 3, 4 * self.h, 4 * self.w), generator=g).to(dev),
                                 torch.randn((self.B, self.K, 3, 4 * self.h, 4 * self.w), generator=g).to(dev)))
        self.results = []
        self.last = None

    def _eager(self, feats_or_images, pyramid, ext, poses, Ks, invK):
        if self.prior:
            return self.model.forward_tensors(feats_or_images[0], feats_or_images[1], ext, poses, Ks, invK,
                                              return_mask=True)
        if self.with_encoder:
            cur_f, src_f = self.model.compute_matching_feats(feats_or_images[0], feats_or_images[1], False)
        else:
            cur_f, src_f = feats_or_images
        return self.model.hot_path(pyramid, cur_f, src_f, ext, poses, Ks, invK, return_mask=True)

    def step(self, i=0):
        inp, cur_image, src_image = self.batches[0 if self.use_graph else i % len(self.batches)]   # (a graph replays on its own buffers)
        first = [cur_image, src_image] if self.with_encoder else [inp["cur_feats"], inp["src_feats"]]
        args = (first, None if self.prior else self.pyramid, inp["src_extrinsics"], inp["src_poses"], inp["src_Ks"],
                inp["cur_invK"])
        if self.use_graph:
            if self._graphed is None:  # capture once (warm-up steps): the ~230 launches of a step become one HIP graph
                from simplerecon_amd.graph import GraphedCallable
                self._graphed = GraphedCallable(self._eager, *args)
                self._static = self._graphed.static_inputs   # inputs live in the graph's buffers: no copies per step
            self.last = self._graphed(*self._static)
        else:
            self.last = self._eager(*args)

    def finish(self, world, force_collective=False):
        # the job's only exchange: the depth maps of the last batch to rank 0 (keyframe i lives on rank i mod world);
        # hero_cfg4_stream gathers EVERY depth map of the run instead
        depth = self.last["depth_pred_s0_b1hw"]
        sharding.gather_results(depth, world * depth.shape[0], dst=0, force_collective=force_collective)

    def config(self, world):
        kind = "FeatureVolumeManager (metadata-MLP matching)" if self.feature_volume_type == "mlp_feature_volume" \
            else "CostVolumeManager (dot-product matching)"
        enc = (f"ResnetMatchingEncoder on {self.B}x{self.K + 1} images -> " if self.with_encoder else "")
        if self.prior:
            enc = (f"EfficientNetV2-S image-prior encoder on {self.B} images (side HIP stream) + " + enc)
        skipped = ("whole DepthModel.forward timed, nothing skipped" if self.prior else
                   "image-prior encoder not timed: its pyramid is a synthetic input" if self.with_encoder else
                   "image-prior and matching encoders not timed: their outputs are synthetic inputs")
        return {"workload": f"{self.name}: hot path = {enc}{kind} -> CVEncoder -> DepthDecoderPP -> exp, batch "
                            f"{self.B}/GPU, {self.K} source views, {self.D} planes, 640x480 image ({self.h}x{self.w} "
                            f"matching features x {self.Cc} ch, image-prior pyramid 24/48/64/160/256 ch), fp32, "
                            f"random-init weights; {skipped}"
                            + (" (BASELINE.json configs[2]: hero_model.yaml, batch 8)" if self.B == 8 and
                               self.feature_volume_type == "mlp_feature_volume" else ""),
                "frames_per_step_per_gpu": self.B,
                "resident_input_batches": 1 if self.use_graph else len(self.batches),   # step i runs on batch i mod this
                # main stream(s) + the image-prior encoder's side stream + the two branch streams of the decoder (its
                # right / diagonal / up branches fork for every node up to `branch_stream_max_regions` regions)
                "hip_streams_per_gpu": self.streams + (1 if (self.prior and getattr(self.model, "prior_on_side_stream", True))
                                                       else 0) + (2 if self._decoder_forks() else 0),
                "submission": "one HIP graph replay per step" if self.use_graph else "eager (one launch per kernel)",
                "parallelism": f"replica x{world} (keyframes sharded)"}

    def _decoder_forks(self):
        dec = getattr(self.model, "depth_decoder", None)
        if dec is None:
            return False
        # the smallest node (15 x 20 at 640 x 480) decides whether any node forks
        regions = self.B * ((self.h // 8 + 7) // 8) * ((self.w // 8 + 15) // 16)
        return self.B <= dec.branch_stream_max_batch or regions <= dec.branch_stream_max_regions

    def _profile_convs(self, n):
        from simplerecon_amd import ops
        inp = self.inp
        with torch.inference_mode():
            volume = lambda: self.model.cost_volume(cur_feats=inp["cur_feats"], src_feats=inp["src_feats"],
                                                    src_extrinsics=inp["src_extrinsics"], src_poses=inp["src_poses"],
                                                    src_Ks=inp["src_Ks"], cur_invK=inp["cur_invK"], min_depth=inp["min_depth"],
                                                    max_depth=inp["max_depth"])[0]
            vol = volume()
            torch.cuda.synchronize()
            ops.PROFILE = []
            try:
                for _ in range(n):
                    pyramid = list(self.model.encoder(self.cur_image)) if self.prior else self.pyramid
                    if self.with_encoder:
                        self.model.compute_matching_feats(self.cur_image, self.src_image, False)
                    vol = volume()   # (r06: the plane sweep is timed IN the step like the convolutions -- VERDICT r05 hygiene (b))
                    feats = self.model.cost_volume_net(vol, pyramid[1:])
                    self.model.depth_decoder(pyramid[:1] + feats)
                torch.cuda.synchronize()
                rec = ops.PROFILE
            finally:
                ops.PROFILE = None
        agg, self._conv_bytes, self._alu_equiv = {}, {}, {}
        for name, flops, e0, e1, shape, executed in rec:
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += flops
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += executed if executed is not None else flops
            if name.startswith("sr_wino4") and executed is not None:
                # `alu_frac`: fp32 MFMA and VALU instructions of a SIMD do not overlap, so the bound that applies
                # is (MFMA clocks + the transforms' minimal VALU clocks) / total.  Per work item and SIMD: S slabs x 144 MFMAs x 32
                # clocks; input transform 72 v_pk_* per slab, output transform 240 per item, 4 clocks each at the vector peak.
                S = (shape[1] + 15) // 16
                self._alu_equiv[name] = self._alu_equiv.get(name, 0.0) + executed * (1.0 + (S * 288.0 + 960.0) / (S * 4608.0))
            if len(shape) >= 10:   # algorithmic bytes of a conv launch: input + output (+ residual) + weights, fp32
                b, ci, h, w, co, k, _s, ho, wo, has_res = shape
                self._conv_bytes[name] = self._conv_bytes.get(name, 0.0) + 4.0 * (
                    b * h * w * ci + b * ho * wo * co * (2 if has_res else 1) + co * ci * k * k + co)
        return agg

    def _mlp_sweep_time(self, n):
        lib = _lib.lib()
        m, inp = self.model.cost_volume, self.inp
        B, K, Cc, h, w, D = self.B, self.K, self.Cc, self.h, self.w, self.D
        planes = m.generate_depth_planes(B, inp["min_depth"], inp["max_depth"])
        vol = torch.empty((B, D, h, w), device=self.dev, memory_format=torch.channels_last)
        ws = torch.empty(lib.sr_mlp_volume_workspace_bytes(B, K, Cc, h, w, 128), dtype=torch.uint8, device=self.dev)
        st = _lib.stream_ptr(self.dev)
        _lib.check(lib.sr_volume_prepare(_lib.ptr(inp["src_feats"]), _lib.ptr(inp["src_Ks"]),
                                         _lib.ptr(inp["src_extrinsics"]), _lib.ptr(inp["src_poses"]), B, K, Cc, h, w,
                                         _lib.ptr(ws), ws.numel(), st), "prepare")
        lin = [mm for mm in m.mlp.net if isinstance(mm, torch.nn.Linear)]
        prm = [t.detach().contiguous() for t in (lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias,
                                                 lin[2].weight, lin[2].bias)]
        _lib.check(lib.sr_mlp_pack_weights(*[_lib.ptr(t) for t in prm], 128, B, K, Cc, h, w, _lib.ptr(ws),
                                           ws.numel(), st), "pack")

        def sweep():
            _lib.check(lib.sr_mlp_volume_sweep(_lib.ptr(inp["cur_feats"]), _lib.ptr(inp["cur_invK"]), _lib.ptr(planes),
                                               *planes.stride(), C.c_float(0.01), B, K, Cc, h, w, D, _lib.ptr(vol),
                                               D * h * w, 1, D, None, None, _lib.ptr(ws), ws.numel(), st), "sweep")
        return _time_launches(sweep, n)

    def _kernel_entries(self, n):
        """One roofline entry per kernel of the step (every conv kernel by name + the plane sweep), sorted by time per step.
        Conv kernels: `achieved` counts the FLOPs the kernel EXECUTES on the matrix cores -- Winograd F(2x2, 3x3) issues 16 and
        F(4x4, 3x3) 36 / 4 = 9 multiplies per 2x2 outputs and channel pair instead of the direct algorithm's 36 -- so frac = MFMA
        utilisation <= 1; the direct-convolution (algorithmic) count is reported beside it."""
        n = max(3, min(n, 10))
        agg = self._profile_convs(n)
        self._prof_n = n
        out = []
        for name, (calls, flops, t, executed) in agg.items():
            wino = "wino" in name
            ex = executed if wino else flops
            peak = FP32_MFMA_PEAK_TF
            if "split" in name:   # fenced experiment: three 16-bit products per fp32 product, priced on the 16-bit matrix pipe
                ex, peak = 3 * ex, F16_MFMA_PEAK_TF
            e = {"kernel": name, "bound": "mfma", "achieved": ex / t / 1e12, "peak": peak, "unit": "TFLOP/s",
                 "frac": ex / t / 1e12 / peak, "avg_launch_us": t / calls * 1e6, "launches_per_step": calls // n,
                 "ms_per_step": t / n * 1e3, "executed_flops_per_launch": ex / calls,
                 "algorithmic_flops_per_launch": flops / calls, "algorithmic_tflops": flops / t / 1e12}
            if name in self._alu_equiv:
                e["alu_frac"] = self._alu_equiv[name] / t / 1e12 / peak
            nbytes = self._conv_bytes.get(name)
            if nbytes:   # both rooflines of the kernel: the larger fraction names the resource that binds it
                e["mfma_frac"] = e["frac"]
                e["hbm_frac"] = nbytes / t / 1e9 / HBM_PEAK_GBS
                e["algorithmic_bytes_per_launch"] = nbytes / calls
                if e["hbm_frac"] > e["mfma_frac"]:
                    e.update({"bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": e["hbm_frac"]})
            out.append(e)
        out = [e for e in out if e["kernel"] != "sr_mlp_volume_fwd"]   # (the sweep's in-step record: its own entry below)
        if self.feature_volume_type == "mlp_feature_volume":
            t_iso = self._mlp_sweep_time(n)
            calls, _, t_in, _ = agg.get("sr_mlp_volume_fwd", (0, 0.0, 0.0, 0.0))
            # in-step time of sr_mlp_volume_fwd = geometry records + weight packing + the sweep launch (one C call); the isolated
            # sweep launch is reported beside it
            t = t_in / calls if calls else t_iso
            N = self.h * self.w
            cin = self.Cc * (self.K + 1) + 10 * self.K + 4
            flops = 2.0 * (cin * 128 + 128 * 128 + 128) * self.B * self.D * N
            e = {"kernel": _mlp_kernel_name(self.K), "bound": "mfma", "achieved": flops / t / 1e12,
                 "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": flops / t / 1e12 / FP32_MFMA_PEAK_TF,
                 "avg_launch_us": t * 1e6, "launches_per_step": 1, "ms_per_step": t * 1e3,
                 "algorithmic_flops_per_launch": flops, "executed_flops_per_launch": flops,
                 "algorithmic_bytes_per_launch": self.B * (4 * ((self.K + 1) * self.Cc * N + self.D * N + N)),
                 "timed": "in the step (sr_mlp_volume_fwd: geometry records + weight packing + the sweep launch)" if calls else
                          "isolated sweep launches",
                 "isolated_sweep_us": t_iso * 1e6}
            if "split" in e["kernel"]:   # fenced experiment: three 16-bit products per fp32 product, priced on the 16-bit pipe
                e.update({"achieved": 3 * flops / t / 1e12, "peak": F16_MFMA_PEAK_TF, "frac": 3 * flops / t / 1e12 / F16_MFMA_PEAK_TF,
                          "algorithmic_tflops": flops / t / 1e12,
                          "note": "executed = 3 x algorithmic flops (two 16-bit pieces per operand, three products)"})
            out.append(e)
        out.sort(key=lambda e: -e["ms_per_step"])
        return out

    def roofline(self, n):
        """The DOMINANT kernel of the step -- the one with the most time per step among every conv kernel and the plane sweep
        (r05: the F(4x4) Winograd kernel and the metadata-MLP sweep are within a few percent of each other; whichever leads is
        named here, the other is the first entry of `kernels`)."""
        ents = self._kernel_entries(n)
        self._entries = ents
        out = dict(ents[0])
        # PMC traffic of THIS kernel (profiles/traffic.json: "<workload>:<kernel symbol>", written by scripts/pmc_traffic.py)
        short = out["kernel"].split("<")[0].split("(")[0].replace("void ", "").strip()
        traffic = _pmc_traffic(f"{self.name}:{short}", out["kernel"])
        out["traffic"] = traffic
        ab = out.get("algorithmic_bytes_per_launch")
        out["traffic_over_algorithmic"] = (traffic / ab) if (traffic and ab) else None
        out["note"] = ("fp32-in / fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32, 157.3 TFLOP/s dense).  "
                       "achieved / frac = FLOPs actually issued on the matrix pipe / time / peak (MFMA utilisation); algorithmic_* = "
                       "the direct-convolution count 2*B*Ho*Wo*Cout*Cin*k*k over the launches of this kernel in one step "
                       "(Winograd issues 16/36 -- F(4x4): 9/36 -- of them, so algorithmic_tflops may exceed the peak: a speed-up "
                       "over direct convolution, not a roofline fraction)")
        return out

    def roofline_hbm(self, n):
        """North star: 'achieved HBM GB/s on the warp/reduce kernel': the fused plane-sweep kernel this workload runs
        (metadata-MLP sweep for the hero model) on its algorithmic bytes, plus the dot-product sweep kernel at the
        same shapes (BASELINE.json configs[1] batched) for comparison."""
        B, K, Cc, h, w, D = self.B, self.K, self.Cc, self.h, self.w, self.D
        N = h * w
        nbytes = B * (4 * ((K + 1) * Cc * N + D * N + N) + 4 * (32 * K + 16 + D))
        out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": nbytes}
        if self.feature_volume_type == "mlp_feature_volume":
            t = self._mlp_sweep_time(max(3, min(n, 10)))
            out.update({"kernel": _mlp_kernel_name(K), "achieved": nbytes / t / 1e9, "frac": nbytes / t / 1e9 / HBM_PEAK_GBS,
                        "avg_launch_us": t * 1e6, "traffic": _pmc_traffic(self.name + ":mlp_sweep"),
                        "note": "the hero model's sweep is MFMA-bound (see kernels[]): its HBM fraction is what the "
                                "compulsory bytes amount to at that speed"})
        dot = DotCfg2(self.dev, 0, B=B, D=D, name=f"dot_b{B}")
        r = dot.roofline(max(n, 10))
        out["dot_sweep"] = {k: r[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "algorithmic_bytes_per_launch",
                                              "traffic", "traffic_over_algorithmic", "issue_lds_bound")}
        if "kernel" not in out:
            out.update(out.pop("dot_sweep"))
        return out

    def extra_kernels(self, n):
        return [dict(e) for e in self._entries[1:]]

    def cpu_baseline(self):
        """The reference's CPU PyTorch path for this workload = the same ATen operator sequence (bench_cpu_aten.py:
        F.grid_sample / F.normalize / torch.cat / F.linear per plane, F.conv2d BasicBlocks, F.interpolate) on all host
        cores; 1 frame (the GPU step is a batch of B), repeated until ~12 s have passed."""
        import copy
        import types
        import bench_cpu_aten as aten
        probe = {}
        threads = _best_cpu_threads(self.K, self.Cc, self.h, self.w, probe)
        m = self.model   # CPU copies of the parameter-holding sub-modules only (no streams / workspaces)
        cpu_model = types.SimpleNamespace(
            encoder=copy.deepcopy(m.encoder).cpu() if self.prior else None,
            matching_model=copy.deepcopy(m.matching_model).cpu() if self.with_encoder else None,
            cost_volume=types.SimpleNamespace(mlp=copy.deepcopy(m.cost_volume.mlp).cpu()
                                              if hasattr(m.cost_volume, "mlp") else None),
            cost_volume_net=copy.deepcopy(m.cost_volume_net).cpu(), depth_decoder=copy.deepcopy(m.depth_decoder).cpu())
        c = {k: (v[:1].cpu() if v.dim() > 0 and v.shape[0] == self.B and k not in ("min_depth", "max_depth") else v.cpu())
             for k, v in self.inp.items()}
        planes = self.model.cost_volume.generate_depth_planes(1, self.inp["min_depth"], self.inp["max_depth"])
        planes = planes[:, :, 0, 0].cpu().contiguous()
        pyr = [f[:1].cpu().contiguous() for f in self.pyramid]
        cur_img = self.cur_image[:1].cpu() if self.with_encoder else None
        src_img = self.src_image[:1].cpu() if self.with_encoder else None
        mlp = self.feature_volume_type == "mlp_feature_volume"

        # Bounded sample (the default bench run must finish within minutes): the plane sweep is timed on SAMPLE planes
        # and scaled to the D planes of the workload (its cost is per plane, the reference loops over planes:
        # cost_volume.py:553); everything else (encoders, CVEncoder, decoder) is timed whole, once.
        SAMPLE = min(16, planes.shape[1])   # (r04: 4 planes -- too few for a number that has to hold from box to box)
        D = planes.shape[1]
        sub = planes[:, torch.linspace(0, D - 1, SAMPLE).round().long()].contiguous()

        def timed(fn, budget):
            with torch.inference_mode():
                fn()                                       # warm-up (oneDNN primitive creation, thread pool)
                n, t0 = 0, time.perf_counter()
                while n == 0 or time.perf_counter() - t0 < budget:
                    fn()
                    n += 1
                return (time.perf_counter() - t0) / n, n

        feats = {}

        def enc():
            if self.prior:
                feats["pyr"] = aten.image_prior_encoder(cpu_model.encoder, cur_img)
            if self.with_encoder:
                f = aten.matching_encoder(cpu_model.matching_model,
                                          torch.cat([cur_img.unsqueeze(1), src_img], 1).flatten(0, 1))
                feats["cur"], feats["src"] = f[:1], f[1:].unsqueeze(0)
        feats["pyr"], feats["cur"], feats["src"] = pyr, c["cur_feats"], c["src_feats"]
        t_enc, n_enc = timed(enc, 2.0) if (self.prior or self.with_encoder) else (0.0, 0)

        def sweep():
            if mlp:
                lin = [(m_.weight, m_.bias) for m_ in cpu_model.cost_volume.mlp.net if isinstance(m_, torch.nn.Linear)]
                feats["vol"] = aten.mlp_volume(feats["cur"], feats["src"], c["src_Ks"], c["src_extrinsics"],
                                               c["src_poses"], c["cur_invK"], sub, lin)[0]
            else:
                feats["vol"] = aten.dot_volume(feats["cur"], feats["src"], c["src_Ks"], c["src_extrinsics"],
                                               c["cur_invK"], sub)[0]
        t_sweep, n_sweep = timed(sweep, 3.0)
        vol_full = torch.randn((1, D, self.h, self.w))    # the conv stack's cost does not depend on the values

        def convs():
            e = aten.cv_encoder(cpu_model.cost_volume_net, vol_full, feats["pyr"][1:])
            aten.depth_decoder(cpu_model.depth_decoder, [feats["pyr"][0]] + e)
        t_conv, n_conv = timed(convs, 3.0)
        dt = t_enc + t_sweep * (D / SAMPLE) + t_conv
        return {"value": 1.0 / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                "thread_probe_seconds_per_plane": probe,
                "seconds_per_frame": {"encoders": t_enc, "plane_sweep": t_sweep * (D / SAMPLE), "conv_stack": t_conv},
                "sample": f"1 frame of {self.name} through bench_cpu_aten.py = the ATen (PyTorch {torch.__version__} CPU, "
                          f"fp32) operator sequence the reference runs on CPU (per-plane F.grid_sample / F.normalize / "
                          f"torch.cat / F.linear of the looped FeatureVolumeManager, F.conv2d BasicBlocks, F.interpolate), "
                          f"torch.set_num_threads({torch.get_num_threads()}) = the fastest of 4..128 probed on two planes of this very sweep "
                          f"(thread_probe_seconds_per_plane; host has {os.cpu_count()} CPUs); bounded sample: "
                          f"plane sweep timed on {SAMPLE} of {D} planes ({n_sweep} repetition(s)) and scaled by {D}/{SAMPLE}, "
                          f"{'encoders (' + str(n_enc) + ' rep) and ' if n_enc else ''}CVEncoder + DepthDecoderPP "
                          f"({n_conv} rep) timed whole"}


def stream_batch_ids(rank, world, batch, max_batches, step):
    """Global keyframe ids of rank `rank`'s batch `step`: its round-robin shard of a stream of world x max_batches x
    batch keyframes (sharding.shard_indices), cut into batches (sharding.batches)."""
    shard = sharding.shard_indices(world * max_batches * batch, rank, world)
    return sharding.batches(shard, batch)[step % max_batches]


class HeroCfg4Stream(HeroCfg3):
    """BASELINE.json configs[3]: hero_model.yaml on a synthetic ScanNet-shaped STREAM of keyframes sharded round-robin
    over the GPUs (keyframe i -> rank i mod world, sharding.shard_indices; reference loop test.py:257-280), batches of 8,
    and the depth map of EVERY keyframe gathered to rank 0 over RCCL (sharding.gather_results) inside the timed region.
    The stream has world x steps x 8 keyframes (2048 = the BASELINE config at 8 GPUs x 32 steps); keyframe i's images are
    generated on the device from a generator seeded with i, its cameras are synthetic.keyframe_poses (the DVMVS-like
    layout with a rigid jitter seeded by i): a keyframe is the same data on whatever rank / in whatever batch it lands,
    so the gathered result of N ranks equals the 1-rank result (tests/test_gpu_multirank.py)."""
    name = "hero_cfg4_stream"
    ROTATE = 1   # (every step generates its own keyframes)

    def __init__(self, dev, rank, max_batches=256):
        super().__init__(dev, rank, name="hero_cfg4_stream")
        self.rank, self.max_batches = rank, max_batches
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        Kmat, invK = synthetic.intrinsics(self.h, self.w)
        self.src_Ks = torch.from_numpy(np.broadcast_to(Kmat, (self.B, self.K, 4, 4)).copy()).to(dev)
        self.cur_invK = torch.from_numpy(np.broadcast_to(invK, (self.B, 4, 4)).copy()).to(dev)
        self._pose_cache = {}
        self._gen = torch.Generator(device=dev)
        self.results = []
        self.gathered = None

    def _batch_ids(self, step):
        return stream_batch_ids(self.rank, self.world, self.B, self.max_batches, step)

    def _poses(self, ids):
        hit = self._pose_cache.get(ids[0])
        if hit is None:
            poses, extr = synthetic.keyframe_poses(ids, self.K)
            hit = (torch.from_numpy(poses).to(self.dev), torch.from_numpy(extr).to(self.dev))
            self._pose_cache[ids[0]] = hit
        return hit

    def keyframes(self, ids):
        """Images and cameras of the keyframes `ids`: functions of the keyframe id alone, so a keyframe is the same
        data whatever rank and batch it lands in (the gathered stream of N ranks equals the 1-rank stream)."""
        cur = torch.empty((len(ids), 3, 4 * self.h, 4 * self.w), device=self.dev)
        src = torch.empty((len(ids), self.K, 3, 4 * self.h, 4 * self.w), device=self.dev)
        for j, i in enumerate(ids):
            g = self._gen.manual_seed(int(i))
            cur[j].normal_(generator=g)
            src[j].normal_(generator=g)
        poses, extr = self._poses(ids)
        return cur, src, poses, extr

    def step(self, i=None):
        if i == 0 or i is None:
            self.results = []          # warm-up steps and the start of the timed region
        ids = self._batch_ids(0 if i is None else i)
        cur, src, poses, extr = self.keyframes(ids)
        out = self.model.forward_tensors(cur, src, extr, poses, self.src_Ks, self.cur_invK, return_mask=True)
        self.last = out
        self.results.append(out["depth_pred_s0_b1hw"])

    def finish(self, world, force_collective=False):
        local = torch.cat(self.results, 0)
        self.gathered = sharding.gather_results(local, world * local.shape[0], dst=0, force_collective=force_collective)

    def config(self, world):
        c = super().config(world)
        c["workload"] = (f"{self.name}: stream of world x steps x {self.B} synthetic keyframes (generated on-device, seeded "
                         f"per keyframe), sharded round-robin over {world} GPU(s) in batches of {self.B}; whole "
                         f"DepthModel.forward per batch (hero_model.yaml: {self.K} source views, {self.D} planes, 640x480, "
                         f"fp32, random-init weights); every depth map gathered to rank 0 inside the timed region "
                         f"(BASELINE.json configs[3]: 2048 keyframes = 8 GPUs x 32 steps)")
        return c

    def cpu_baseline(self):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port",
                "sample": "not run for this workload (same per-frame work as hero_cfg3)"}


class HeroVolumeOnly:
    """FeatureVolumeManager alone (metadata-MLP sweep); default shape = BASELINE.json configs[4], the stress
    config: 15 source views, 96 planes, 960x720 (matching 180x240), batch 4.  The UNet++ cannot run at
    960x720 (reference bug: 23 -> 46 != 45 rows), so this config is the cost volume only."""
    name = "hero_cfg5_volume"

    def __init__(self, dev, rank, B=4, K=15, D=96, h=180, w=240):
        from simplerecon_amd.cost_volume import FeatureVolumeManager
        self.B, self.K, self.D, self.h, self.w, self.Cc = B, K, D, h, w, 16
        self.dev = dev
        self.frames_per_step = B
        self.inp = synthetic.cost_volume_inputs(B, K, 16, h, w, seed=rank, device=dev)
        mgr = FeatureVolumeManager(h, w, num_depth_bins=D, matching_dim_size=16, num_source_views=K)
        synthetic.seeded_fill_(mgr.mlp, seed=3)
        self.mgr = mgr.to(dev)
        self.mgr.volume_memory_format = torch.channels_last
        self.last = None

    def step(self, i=0):
        self.last = self.mgr(return_mask=True, **self.inp)

    def finish(self, world, force_collective=False):
        # the job's only exchange: this step's results to rank 0 (keyframe i lives on rank i mod world)
        sharding.gather_results(self.last[1], world * self.last[1].shape[0], dst=0, force_collective=force_collective)

    def config(self, world):
        return {"workload": f"{self.name}: FeatureVolumeManager (metadata-MLP sweep) only, batch {self.B}/GPU, {self.K} "
                            f"source views, {self.D} planes, {4*self.w}x{4*self.h} image -> {self.h}x{self.w} matching "
                            f"features x 16 ch, fp32 (BASELINE.json configs[4], cost volume part)",
                "frames_per_step_per_gpu": self.B, "parallelism": f"replica x{world} (keyframes sharded)"}

    def roofline(self, n):
        t = _time_launches(lambda: self.step(), max(3, min(n, 10)))
        N = self.h * self.w
        cin = 16 * (self.K + 1) + 10 * self.K + 4
        flops = 2.0 * (cin * 128 + 128 * 128 + 128) * self.B * self.D * N
        name = _mlp_kernel_name(self.K)
        return {"kernel": name, "bound": "mfma", "achieved": flops / t / 1e12, "peak": FP32_MFMA_PEAK_TF,
                "unit": "TFLOP/s", "frac": flops / t / 1e12 / FP32_MFMA_PEAK_TF, "traffic": _pmc_traffic(self.name),
                "avg_launch_us": t * 1e6, "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": self.B * 4 * ((self.K + 1) * 16 * N + self.D * N + N),
                "note": "time = whole FeatureVolumeManager.forward (prepare + pack + sweep + argmax), sweep dominates"}

    def extra_kernels(self, n):
        return None

    def cpu_baseline(self):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "not run for this workload"}


class TsdfFuse:
    """TSDF fusion of predicted depth maps ("next" component after the hot path, reference tools/tsdf.py +
    fusers_helper.OurFuser): batches of 8 640x480 fp16 depth maps into the reference's default +-10 m cube at 4 cm
    (504^3 = 128 M voxels, fp16 values + weights = 512 MB), camera moving through the volume."""
    name = "tsdf_fuse"

    def __init__(self, dev, rank, B=8, bounds=None):
        from simplerecon_amd.tsdf import OurFuser
        self.dev, self.B = dev, B
        self.frames_per_step = B
        self.fuser = OurFuser(bounds=bounds, max_fusion_depth=3.0, device=dev)
        g = torch.Generator(device="cpu").manual_seed(2000 + rank)
        self.depth = (1.0 + 1.5 * torch.rand((B, 1, 480, 640), generator=g)).to(dev).half()
        K = torch.eye(4).repeat(B, 1, 1)
        K[:, 0, 0] = K[:, 1, 1] = 577.87
        K[:, 0, 2], K[:, 1, 2] = 320.0, 240.0
        self.K = K.to(dev).half()
        self.step_i = 0

    def _poses(self, i):
        T = torch.eye(4).repeat(self.B, 1, 1)
        for k in range(self.B):   # a slow sweep: 5 cm per frame along x, small yaw
            t = 0.05 * (i * self.B + k)
            a = 0.02 * (i * self.B + k)
            T[k, 0, 0], T[k, 0, 2], T[k, 2, 0], T[k, 2, 2] = np.cos(a), np.sin(a), -np.sin(a), np.cos(a)
            T[k, 0, 3] = -t % 4.0
        return T.to(self.dev).half()

    def step(self, i=0):
        self.step_i += 1
        self.fuser.tsdf_fuser_pred.integrate_depth(self.depth, self._poses(self.step_i), self.K)

    def finish(self, world, force_collective=False):
        pass  # each rank fuses its own scene (scenes are independent; a volume is never split across GPUs)

    def config(self, world):
        f = self.fuser.tsdf_fuser_pred
        return {"workload": f"{self.name}: TSDFFuser.integrate_depth, batch of {self.B} 640x480 fp16 depth maps into a "
                            f"{'x'.join(str(int(d)) for d in f.shape)} fp16 volume at {f.voxel_size} m "
                            f"(reference default +-10 m bounds), max depth 3 m",
                "frames_per_step_per_gpu": self.B, "parallelism": f"replica x{world} (one scene per GPU)"}

    def roofline(self, n):
        f = self.fuser.tsdf_fuser_pred
        T = self._poses(1)
        t = _time_launches(lambda: f.integrate_depth(self.depth, T, self.K), max(5, n))
        vox = f.tsdf_values.numel()
        nbytes = 4.0 * vox + 2.0 * self.depth.numel()
        return {"kernel": "sr_tsdf_integrate_kernel", "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS, "traffic": _pmc_traffic(self.name),
                "avg_launch_us": t * 1e6, "algorithmic_bytes_per_launch": nbytes,
                "note": "algorithmic bytes = one read of the fp16 values + weights of every voxel + the depth maps; the "
                        "kernel rejects tiles / segments outside the view frusta on their corners WITHOUT reading them, "
                        "so it is bound by that rejection arithmetic (VALU), not by HBM: real traffic is far below"}

    def extra_kernels(self, n):
        return None

    def cpu_baseline(self):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "not run for this workload"}


class HeroCfg5(HeroCfg3):
    """BASELINE.json configs[4] end to end: hero_model.yaml stress configuration -- 15 source views, 96 planes, batch 4 -- at
    960x736 (the reference's UNet++ cannot run at 960x720: 23 -> 46 != 45 rows, modules/networks.py:83-89; 736 = 23 * 32 is
    SURVEY.md §7's padded variant).  Whole DepthModel.forward from the images; parity at this shape:
    tests/test_gpu_e2e_stress_size.py."""
    name = "hero_cfg5"
    B, K, Cc, D, h, w = 4, 15, 16, 96, 184, 240

    def config(self, world):
        c = super().config(world)
        c["workload"] = (f"{self.name}: whole DepthModel.forward (EfficientNetV2-S image-prior encoder + ResnetMatchingEncoder on "
                         f"{self.B}x{self.K + 1} images -> 410-input metadata-MLP sweep -> CVEncoder -> DepthDecoderPP -> exp), batch "
                         f"{self.B}/GPU, {self.K} source views, {self.D} planes, 960x736 image ({self.h}x{self.w} matching "
                         f"features), fp32, random-init weights (BASELINE.json configs[4] padded from 960x720, where the "
                         f"reference's decoder cannot run)")
        return c

    def cpu_baseline(self):
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "not run for this workload"}


class DotFull(HeroCfg3):
    """dot_product_model.yaml through the full hot path (cost volume + conv stack), batch 8."""
    name = "dot_full"
    feature_volume_type = "simple_cost_volume"


WORKLOADS = {
    "hero_cfg3": lambda dev, rank: HeroCfg3(dev, rank),
    "hero_b1": lambda dev, rank: HeroCfg3(dev, rank, B=1, name="hero_b1"),
    "hero_cfg3_noprior": lambda dev, rank: HeroCfg3(dev, rank, prior=False, name="hero_cfg3_noprior"),
    "hero_b1_noprior": lambda dev, rank: HeroCfg3(dev, rank, B=1, prior=False, name="hero_b1_noprior"),
    "hero_cfg3_core": lambda dev, rank: HeroCfg3(dev, rank, with_encoder=False, name="hero_cfg3_core"),
    "hero_b1_core": lambda dev, rank: HeroCfg3(dev, rank, B=1, with_encoder=False, name="hero_b1_core"),
    "hero_cfg3_graph": lambda dev, rank: HeroCfg3(dev, rank, graph=True, name="hero_cfg3_graph"),
    # fenced experiments: split-precision MLP sweep; everything else as hero_cfg3
    "hero_cfg3_bf16x3": lambda dev, rank: HeroCfg3(dev, rank, split="bf16", name="hero_cfg3_bf16x3"),
    "hero_cfg3_f16x3": lambda dev, rank: HeroCfg3(dev, rank, split="f16", name="hero_cfg3_f16x3"),
    "hero_cfg3_bf16x3_convs": lambda dev, rank: HeroCfg3(dev, rank, split="bf16", split_convs=True, name="hero_cfg3_bf16x3_convs"),
    "hero_cfg3_f16x3_convs": lambda dev, rank: HeroCfg3(dev, rank, split="f16", split_convs=True, name="hero_cfg3_f16x3_convs"),
    "hero_b1_graph_f16x3_convs": lambda dev, rank: HeroCfg3(dev, rank, B=1, graph=True, split="f16", split_convs=True,
                                                            name="hero_b1_graph_f16x3_convs"),
    "hero_b1_graph": lambda dev, rank: HeroCfg3(dev, rank, B=1, graph=True, name="hero_b1_graph"),
    "hero_b1_noprior_graph": lambda dev, rank: HeroCfg3(dev, rank, B=1, prior=False, graph=True, name="hero_b1_noprior_graph"),
    "hero_b1_core_graph": lambda dev, rank: HeroCfg3(dev, rank, B=1, with_encoder=False, graph=True,
                                                     name="hero_b1_core_graph"),
    "hero_cfg3_s2": lambda dev, rank: HeroCfg3(dev, rank, streams=2),
    "hero_cfg3_s4": lambda dev, rank: HeroCfg3(dev, rank, streams=4),
    "dot_full": lambda dev, rank: DotFull(dev, rank),
    "hero_cfg4_stream": lambda dev, rank: HeroCfg4Stream(dev, rank),
    "hero_cfg5_volume": lambda dev, rank: HeroVolumeOnly(dev, rank),
    "hero_cfg5": lambda dev, rank: HeroCfg5(dev, rank),
    "hero_cfg3_volume": lambda dev, rank: HeroVolumeOnly(dev, rank, B=8, K=7, D=64, h=120, w=160),
    "tsdf_fuse": lambda dev, rank: TsdfFuse(dev, rank),
    "dot_cfg2": lambda dev, rank: DotCfg2(dev, rank),
    "dot_b8": lambda dev, rank: DotCfg2(dev, rank, B=8, name="dot_b8"),
    # evidence for "the sweep is HBM-bound only when there are few planes": 2 planes, batch 64 (not a BASELINE config)
    "dot_d2_b64": lambda dev, rank: DotCfg2(dev, rank, B=64, D=2, name="dot_d2_b64"),
}
DEFAULT = "hero_cfg3"
