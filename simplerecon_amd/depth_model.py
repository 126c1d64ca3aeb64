"""DepthModel shell -- the caller of the hot path, API-compatible with the reference's
experiment_modules/depth_model.py `DepthModel.forward` (:247-407) and `__init__` (:68-189),
without the PyTorch-Lightning / losses / training machinery (out of scope).

    model = DepthModel(opts)                      # opts: reference options.Options or default_options()
    outputs = model("test", cur_data, src_data, unbatched_matching_encoder_forward=True, return_mask=True)

Both encoders are built natively on the HIP kernels: the matching encoder (antialiased ResNet-18 stem +
InstanceNorm tail, reference networks.py:149-205) = networks.ResnetMatchingEncoder, and
the image-prior encoder (timm tf_efficientnetv2_s feature pyramid, reference depth_model.py:110-116,
SURVEY.md §8f "next" #1) = image_encoder.EfficientNetV2SFeatures (third-party architecture restated from
its public definition; timm's parameter names, so `encoder.*` checkpoint entries load).  Both stay
pluggable (`image_encoder=`, `matching_encoder=`); `timm_image_encoder()` builds the reference's own
constructor call when timm is importable.  `hot_path()` starts from the encoders' outputs.
"""
import contextlib
import os
from dataclasses import dataclass

import torch
from torch import nn

from .cost_volume import CostVolumeManager, FeatureVolumeManager
from .layers import TensorFormatter
from . import autograd_ops, ops
from .image_encoder import EfficientNetV2SFeatures
from .networks import CVEncoder, DepthDecoderPP, ResnetMatchingEncoder


@dataclass
class Options:
    """The subset of the reference's options.Options (options.py:9-216) this path reads."""
    image_encoder_name: str = "efficientnet"
    cv_encoder_type: str = "multi_scale_encoder"
    depth_decoder_name: str = "unet_pp"
    matching_encoder_type: str = "resnet"
    feature_volume_type: str = "mlp_feature_volume"   # hero_model.yaml; "simple_cost_volume" = dot_product_model.yaml
    matching_num_depth_bins: int = 64
    matching_scale: int = 1
    matching_feature_dims: int = 16
    model_num_views: int = 8
    image_width: int = 512
    image_height: int = 384
    min_matching_depth: float = 0.25
    max_matching_depth: float = 5.0
    loss_type: str = "log_l1"


def default_options(**kw):
    return Options(**kw)


IMAGE_PRIOR_CHANNELS = [24, 48, 64, 160, 256]  # tf_efficientnetv2_s features_only (depth_model.py:110-118)


class PendingPyramid:
    """The image-prior pyramid while its encoder is still running on a side HIP stream.  The pyramid is first needed
    by the CVEncoder, AFTER the matching encoder and the plane sweep: `wait()` joins the side stream into the
    current one at that point, so the encoder's many small, latency-bound launches overlap the big kernels of the
    main stream.  (The side stream always forks from the main stream first -- `side.wait_stream(main)` -- which also
    orders the allocator's reuse of the pyramid buffers behind the previous frame's consumers.)"""

    def __init__(self, stream, feats, events=None):
        self.stream, self.feats = stream, list(feats)
        self.events = events   # one per pyramid level, recorded on `stream` right behind the launch that wrote the level

    def wait(self):
        torch.cuda.current_stream(self.feats[0].device).wait_stream(self.stream)
        return self.feats

    def levels_from(self, first):
        """Levels first.. as a sequence whose items join only as much of the encoder as produced them (r06): at batch 1
        the encoder's ~200 small launches outlast matching encoder + sweep, and the first CVEncoder levels need only the
        shallow pyramid levels -- they run while stages 5 / 6 of the encoder are still in flight."""
        return _PendingLevels(self, first)


class _PendingLevels:
    def __init__(self, pending, first):
        self.pending, self.first = pending, first

    def __len__(self):
        return len(self.pending.feats) - self.first

    def peek(self):
        """The tensors without any stream join (shapes / requires_grad checks only -- not for launching work on them)."""
        return self.pending.feats[self.first:]

    def __getitem__(self, i):
        p = self.pending
        if isinstance(i, slice) or i < 0:
            raise TypeError("pending pyramid levels are taken one at a time, by non-negative index")
        lvl = self.first + i
        if p.events is None:
            p.wait()
        else:
            torch.cuda.current_stream(p.feats[lvl].device).wait_event(p.events[lvl])
        return p.feats[lvl]


def tensor_B_to_bM(t, batch_size, num_views):
    return t.view([batch_size, num_views] + list(t.shape[1:]))  # reference generic_utils.py:110-118


def tensor_bM_to_B(t):
    return t.view([t.shape[0] * t.shape[1]] + list(t.shape[2:]))  # reference generic_utils.py:121-130


def timm_image_encoder(pretrained=True):
    """The reference's own constructor call (depth_model.py:110-116) -- a torch module, NOT the HIP path; for
    side-by-side checks on machines that have timm."""
    try:
        import timm
    except ImportError as e:
        raise ImportError("timm is not installed; DepthModel's default image-prior encoder is the native "
                          "image_encoder.EfficientNetV2SFeatures") from e
    enc = timm.create_model("tf_efficientnetv2_s_in21ft1k", pretrained=pretrained, features_only=True)
    enc.num_ch_enc = enc.feature_info.channels()
    return enc


class DepthModel(nn.Module):

    def __init__(self, opts, image_encoder=None, matching_encoder=None):
        super().__init__()
        self.run_opts = opts
        if image_encoder is None:
            if "efficientnet" not in opts.image_encoder_name:
                raise ValueError("Unrecognized option for image encoder type!")
            image_encoder = EfficientNetV2SFeatures()
        self.encoder = image_encoder
        num_ch_enc = list(getattr(self.encoder, "num_ch_enc", IMAGE_PRIOR_CHANNELS))

        if opts.cv_encoder_type != "multi_scale_encoder":
            raise ValueError("Unrecognized option for cost volume encoder type!")
        self.cost_volume_net = CVEncoder(num_ch_cv=opts.matching_num_depth_bins,
                                         num_ch_enc=num_ch_enc[opts.matching_scale:],
                                         num_ch_outs=[64, 128, 256, 384])
        dec_in = num_ch_enc[:opts.matching_scale] + self.cost_volume_net.num_ch_enc
        if opts.depth_decoder_name != "unet_pp":
            raise ValueError("Unrecognized option for depth decoder name!")
        self.depth_decoder = DepthDecoderPP(dec_in)

        if opts.feature_volume_type == "simple_cost_volume":
            cost_volume_class = CostVolumeManager
        elif opts.feature_volume_type == "mlp_feature_volume":
            cost_volume_class = FeatureVolumeManager
        else:
            raise ValueError(f"Unrecognized option {opts.feature_volume_type} for feature volume type!")
        self.cost_volume = cost_volume_class(
            matching_height=opts.image_height // (2 ** (opts.matching_scale + 1)),
            matching_width=opts.image_width // (2 ** (opts.matching_scale + 1)),
            num_depth_bins=opts.matching_num_depth_bins,
            matching_dim_size=opts.matching_feature_dims,
            num_source_views=opts.model_num_views - 1,
        )
        # the HIP CVEncoder consumes the volume channels-last: have the sweep write it that way
        self.cost_volume.volume_memory_format = torch.channels_last

        if matching_encoder is None:
            if opts.matching_encoder_type == "resnet":
                # reference depth_model.py:181-182: ResnetMatchingEncoder(18, matching_feature_dims)
                matching_encoder = ResnetMatchingEncoder(18, opts.matching_feature_dims).eval()
            elif opts.matching_encoder_type == "unet_encoder":
                raise NotImplementedError("UNetMatchingEncoder needs timm + torchvision FPN (reference "
                                          "networks.py:207-251); pass matching_encoder=... to plug one in")
            else:
                raise ValueError(f"Unrecognized option {opts.matching_encoder_type} for matching encoder type!")
        self.matching_model = matching_encoder
        self.tensor_formatter = TensorFormatter()
        # Keyframes of a batch are independent: hot_path() can run `num_streams` sub-batches on separate
        # HIP streams so that one sub-batch's kernel tails / launch gaps are filled by the other's work.
        self.num_streams = 1
        self._streams = {}
        self._range_cache = {}
        # run the image-prior encoder on a side HIP stream, concurrently with matching encoder + plane sweep
        self.prior_on_side_stream = True
        self.sweep_reserved_cus = int(os.environ.get("SR_SWEEP_RESERVED_CUS", "32"))   # see _hot_path_one
        self.sweep_reserve_max_points = int(os.environ.get("SR_SWEEP_RESERVE_MAX_POINTS", "5000000"))   # batch 4 at 120x160x64
        self._prior_streams = {}
        # opt-in: run both encoders under no_grad when autograd is recording (frozen-encoder fine-tune)
        self.freeze_encoders = False

    # ---- reference depth_model.py:191-245 ----------------------------------------------------
    def compute_matching_feats(self, cur_image, src_image, unbatched_matching_encoder_forward):
        if not unbatched_matching_encoder_forward and hasattr(self.matching_model, "forward_pair"):
            return self.matching_model.forward_pair(cur_image, src_image)   # no image concatenation
        all_frames = torch.cat([cur_image.unsqueeze(1), src_image], dim=1)
        batch_size, num_views = all_frames.shape[:2]
        if unbatched_matching_encoder_forward:
            feats = torch.cat([self.matching_model(f) for f in tensor_bM_to_B(all_frames).split(1, dim=0)], dim=0)
            feats = tensor_B_to_bM(feats, batch_size=batch_size, num_views=num_views)
        else:
            feats = self.tensor_formatter(all_frames, apply_func=self.matching_model)
        return feats[:, 0], feats[:, 1:].contiguous()

    # ---- the hot path (reference depth_model.py:358-405) -------------------------------------
    def hot_path(self, cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam, cur_cam_T_src_cam,
                 src_K, cur_invK, return_mask=False, flip=False):
        """Everything between the encoders and the output dict: cost volume -> CVEncoder ->
        DepthDecoderPP -> exp.  `cur_feats` is the image-prior pyramid (list of 5)."""
        b = matching_cur_feats.shape[0]
        n = min(self.num_streams, b)
        if n > 1 and isinstance(cur_feats, PendingPyramid):
            cur_feats = cur_feats.wait()
        if n <= 1:
            return self._hot_path_one(cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                                      cur_cam_T_src_cam, src_K, cur_invK, return_mask, flip)
        dev = matching_cur_feats.device
        streams = self._streams.get(dev)
        if streams is None or len(streams) < n:
            streams = [torch.cuda.Stream(device=dev) for _ in range(n)]
            self._streams[dev] = streams
        main = torch.cuda.current_stream(dev)
        bounds = [(b * i) // n for i in range(n + 1)]
        parts = []
        for i in range(n):
            lo, hi = bounds[i], bounds[i + 1]
            st = streams[i]
            st.wait_stream(main)
            with torch.cuda.stream(st):
                parts.append(self._hot_path_one([f[lo:hi] for f in cur_feats], matching_cur_feats[lo:hi],
                                                matching_src_feats[lo:hi], src_cam_T_cur_cam[lo:hi],
                                                cur_cam_T_src_cam[lo:hi], src_K[lo:hi], cur_invK[lo:hi],
                                                return_mask, flip))
        for st in streams[:n]:
            main.wait_stream(st)
        out = {}
        for k in parts[0]:
            vals = [p[k] for p in parts]
            if vals[0] is None:
                out[k] = None
            else:
                for v in vals:
                    v.record_stream(main)
                out[k] = torch.cat(vals, 0)
        return out

    def _depth_range(self, like):
        """min / max matching depth as [1,1,1,1] device tensors (reference depth_model.py:358-359), created once per
        device and dtype so that the hot path issues no host-to-device copy (and can be captured in a HIP graph)."""
        key = (like.device, like.dtype)
        hit = self._range_cache.get(key)
        o = self.run_opts
        if hit is None or hit[0] != (o.min_matching_depth, o.max_matching_depth):
            mn = torch.tensor(o.min_matching_depth).type_as(like).view(1, 1, 1, 1)
            mx = torch.tensor(o.max_matching_depth).type_as(like).view(1, 1, 1, 1)
            hit = ((o.min_matching_depth, o.max_matching_depth), mn, mx)
            self._range_cache[key] = hit
        return hit[1], hit[2]

    def graphed(self, cur_image, src_image, cur_feats, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK,
                return_mask=False):
        """HIP-graph version of `forward_tensors` for fixed shapes: returns a `graph.GraphedCallable` taking
        (cur_image, src_image, cur_feats, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK).  cur_feats=None:
        the image-prior encoder runs inside the graph (the whole model, ~430 launches, in one submission per keyframe
        batch); cur_feats = a pyramid list: it is an input and only matching encoder -> hot_path are captured."""
        from .graph import GraphedCallable

        def step(cur_image, src_image, cur_feats, src_T_cur, cur_T_src, src_K, cur_invK):
            if cur_feats is None:
                return self.forward_tensors(cur_image, src_image, src_T_cur, cur_T_src, src_K, cur_invK,
                                            return_mask=return_mask)
            mc, ms = self.compute_matching_feats(cur_image, src_image, False)
            return self.hot_path(list(cur_feats), mc, ms, src_T_cur, cur_T_src, src_K, cur_invK,
                                 return_mask=return_mask)
        return GraphedCallable(step, cur_image, src_image, None if cur_feats is None else list(cur_feats),
                               src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK)

    def _hot_path_one(self, cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam, cur_cam_T_src_cam,
                      src_K, cur_invK, return_mask=False, flip=False):
        o = self.run_opts
        min_depth, max_depth = self._depth_range(src_K)
        # While the image-prior encoder is still running on its side stream the sweep leaves it `sweep_reserved_cus` CUs: the
        # sweep's persistent workgroups own their CUs, and the encoder's ~170 small launches would be parked for all of it
        # -- for sweeps of up to `sweep_reserve_max_points` (pixel, plane) points, where 14 % more sweep time is less than what the
        # encoder's progress is worth (measured, `profiles/r06_layer_tables.txt`, HIP-graph replays at 640x480 / 64 planes: batch 1
        # 5.75 -> 5.44 ms, batch 2 8.75 -> 8.26, batch 4 14.21 -> 13.78; 32 of 256 CUs is the optimum, 16 / 24 lose; at batch 8 the
        # longer sweep costs what the overlap gains, at 960x736 / 96 planes / batch 4 it loses 2.4 ms).
        reserve = self.sweep_reserved_cus if isinstance(cur_feats, PendingPyramid) and hasattr(self.cost_volume, "_reserve_cus") \
            and matching_cur_feats.shape[0] * matching_cur_feats.shape[2] * matching_cur_feats.shape[3] \
            * getattr(self.cost_volume, "num_depth_bins", 1 << 30) <= self.sweep_reserve_max_points else 0
        if reserve:
            self.cost_volume._reserve_cus = reserve
        try:
            cost_volume, lowest_cost, _, overall_mask_bhw = self.cost_volume(
                cur_feats=matching_cur_feats, src_feats=matching_src_feats, src_extrinsics=src_cam_T_cur_cam,
                src_poses=cur_cam_T_src_cam, src_Ks=src_K, cur_invK=cur_invK, min_depth=min_depth,
                max_depth=max_depth, return_mask=return_mask)
        finally:
            if reserve:
                self.cost_volume._reserve_cus = 0
        if flip:
            cost_volume = torch.flip(cost_volume, (-1,))
        if isinstance(cur_feats, PendingPyramid) and isinstance(self.cost_volume_net, CVEncoder) \
                and isinstance(self.depth_decoder, DepthDecoderPP):
            # CVEncoder level i joins the encoder's side stream only as far as pyramid level matching_scale + i, and its LAST level
            # is handed to the decoder as a closure: the decoder first launches the branches of its first column that do not read
            # it (the full-resolution convolutions among them), then calls it.  The full join (which a HIP-graph capture and the
            # allocator both need) comes last, when the deepest level has been waited for anyway.
            pending = cur_feats
            cost_volume_features, last_level = self.cost_volume_net(cost_volume, pending.levels_from(o.matching_scale),
                                                                    defer_last=True)
            shallow = pending.levels_from(0)
            feats = [shallow[k] for k in range(o.matching_scale)] + cost_volume_features
            depth_outputs = self.depth_decoder(feats, last_input=last_level)
            pending.wait()
        else:
            if isinstance(cur_feats, PendingPyramid):
                cur_feats = cur_feats.wait()
            cost_volume_features = self.cost_volume_net(cost_volume, cur_feats[o.matching_scale:])
            feats = list(cur_feats[:o.matching_scale]) + cost_volume_features
            depth_outputs = self.depth_decoder(feats)
        for k in list(depth_outputs.keys()):
            log_depth = depth_outputs[k].float()
            if flip:
                log_depth = torch.flip(log_depth, (-1,))
            depth_outputs[k] = log_depth
            depth_outputs[k.replace("log_", "")] = autograd_ops.exp(log_depth) if log_depth.requires_grad \
                else ops.exp(log_depth)
        depth_outputs["lowest_cost_bhw"] = lowest_cost
        depth_outputs["overall_mask_bhw"] = overall_mask_bhw
        return depth_outputs

    # ---- reference depth_model.py:247-407 ----------------------------------------------------
    def forward(self, phase, cur_data, src_data, unbatched_matching_encoder_forward=False, return_mask=False):
        o = self.run_opts
        cur_image = cur_data["image_b3hw"]
        src_image = src_data["image_b3hw"]
        src_K = src_data[f"K_s{o.matching_scale}_b44"]
        cur_invK = cur_data[f"invK_s{o.matching_scale}_b44"]
        with torch.autocast("cuda", enabled=False):
            src_cam_T_cur_cam = src_data["cam_T_world_b44"] @ cur_data["world_T_cam_b44"].unsqueeze(1)
            cur_cam_T_src_cam = cur_data["cam_T_world_b44"].unsqueeze(1) @ src_data["world_T_cam_b44"]
        flip = torch.rand(1).item() < (0.5 if phase == "train" else 0.0)
        if flip:
            cur_image = torch.flip(cur_image, (-1,))
            src_image = torch.flip(src_image, (-1,))
        return self.forward_tensors(cur_image, src_image, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK,
                                    unbatched_matching_encoder_forward=unbatched_matching_encoder_forward,
                                    return_mask=return_mask, flip=flip)

    def image_prior_pyramid(self, cur_image):
        """`self.encoder(cur_image)` (reference depth_model.py:358), launched on a side HIP stream when the encoder
        runs HIP kernels on the GPU: returns a PendingPyramid that hot_path() joins where the pyramid is consumed."""
        if not (self.prior_on_side_stream and cur_image.is_cuda and isinstance(self.encoder, EfficientNetV2SFeatures)) \
                or self.encoder._train_path(cur_image):   # the differentiable path stays on the caller's stream
            return list(self.encoder(cur_image))
        dev = cur_image.device
        side = self._prior_streams.get(dev)
        if side is None:
            side = self._prior_streams[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        events = []

        def on_level(_x):
            ev = torch.cuda.Event()
            ev.record(side)
            events.append(ev)
        with torch.cuda.stream(side):
            feats = self.encoder(cur_image, on_level=on_level)
        return PendingPyramid(side, feats, events if len(events) == len(feats) else None)

    def forward_tensors(self, cur_image, src_image, src_cam_T_cur_cam, cur_cam_T_src_cam, src_K, cur_invK,
                        unbatched_matching_encoder_forward=False, return_mask=False, flip=False):
        """DepthModel.forward after the dict unpacking / relative-pose step (reference depth_model.py:358-405):
        image-prior encoder, matching encoder, cost volume, CVEncoder, decoder, exp -- every stage on HIP kernels."""
        # Training (grad mode): every stage is differentiable on HIP kernels -- the two encoders through train_ops
        # (BatchNorm per its own mode), the cost volume, CVEncoder and DepthDecoderPP through autograd_ops -- like the
        # reference's train.py.  `freeze_encoders = True` is the explicit opt-in for a frozen-encoder fine-tune: both
        # encoders then run under no_grad and their outputs enter the graph as constants.
        frozen = torch.is_grad_enabled() and self.freeze_encoders
        with torch.no_grad() if frozen else contextlib.nullcontext():
            cur_feats = self.image_prior_pyramid(cur_image)
            matching_cur_feats, matching_src_feats = self.compute_matching_feats(
                cur_image, src_image, unbatched_matching_encoder_forward)
        if flip:
            matching_cur_feats = torch.flip(matching_cur_feats, (-1,))
            matching_src_feats = torch.flip(matching_src_feats, (-1,))
        return self.hot_path(cur_feats, matching_cur_feats, matching_src_feats, src_cam_T_cur_cam,
                             cur_cam_T_src_cam, src_K, cur_invK, return_mask=return_mask, flip=flip)
