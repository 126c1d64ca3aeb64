"""2-D networks of the hot path -- API/state-dict compatible with reference modules/networks.py:
`MLP` (:129-147), `CVEncoder` (:99-127), `DepthDecoderPP` (:20-96).

Activations flow between our layers as channels_last (NHWC-in-memory) torch tensors -- logically
still b,c,h,w, so callers see the reference's shapes -- and every BasicBlock writes straight into
its slice of the next concat buffer (no torch.cat copies)."""
import os

import numpy as np
import torch
from torch import nn

from .layers import BasicBlock


def double_basic_block(num_ch_in, num_ch_out, num_repeats=2):
    """Sequential(0: BasicBlock, conv_0: BasicBlock, ...) -- same child names as reference networks.py:13-17."""
    layers = nn.Sequential(BasicBlock(num_ch_in, num_ch_out))
    for i in range(num_repeats - 1):
        layers.add_module(f"conv_{i}", BasicBlock(num_ch_out, num_ch_out))
    return layers


class MLP(nn.Module):
    """Linear / LeakyReLU(0.01) stack (reference networks.py:129-147).  Inside the feature volume the
    three layers are fused into the HIP sweep kernel (cost_volume.FeatureVolumeManager reads
    `net.{0,2,4}.{weight,bias}`); calling the module directly applies the layers as written."""

    def __init__(self, channel_list, disable_final_activation=False):
        super().__init__()
        layer_list = []
        for i in range(len(channel_list) - 1):
            layer_list.append(nn.Linear(channel_list[i], channel_list[i + 1]))
            layer_list.append(nn.LeakyReLU(inplace=True))
        if disable_final_activation:
            layer_list = layer_list[:-1]
        self.net = nn.Sequential(*layer_list)

    def forward(self, x):
        """Applies the stack over the last dimension.  Device fp32 inputs run on the HIP 1x1 kernel with
        the LeakyReLU fused into each layer's epilogue."""
        from . import ops
        mods = list(self.net)
        i = 0
        while i < len(mods):
            lin = mods[i]
            act = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU) else None
            x = ops.linear(x, lin, leaky=act.negative_slope if act is not None else None)
            i += 2 if act is not None else 1
        return x


class CVEncoder(nn.Module):
    """Cost-volume + image-prior multi-scale encoder (reference networks.py:99-127)."""

    def __init__(self, num_ch_cv, num_ch_enc, num_ch_outs):
        super().__init__()
        self.convs = nn.ModuleDict()
        self.num_ch_enc = []
        self.num_blocks = len(num_ch_outs)
        self._img_ch = list(num_ch_enc)
        for i in range(self.num_blocks):
            num_ch_in = num_ch_cv if i == 0 else num_ch_outs[i - 1]
            num_ch_out = num_ch_outs[i]
            self.convs[f"ds_conv_{i}"] = BasicBlock(num_ch_in, num_ch_out, stride=1 if i == 0 else 2)
            self.convs[f"conv_{i}"] = nn.Sequential(
                BasicBlock(num_ch_enc[i] + num_ch_out, num_ch_out, stride=1),
                BasicBlock(num_ch_out, num_ch_out, stride=1),
            )
            self.num_ch_enc.append(num_ch_out)

    def _forward_train(self, x, img_feats):
        """Differentiable forward (reference networks.py:120-127 as written: concat with torch.cat, which autograd
        splits again on the way back); every conv runs autograd_ops (HIP forward + backward)."""
        outputs = []
        for i in range(self.num_blocks):
            x = self.convs[f"ds_conv_{i}"](x)
            x = self.convs[f"conv_{i}"](torch.cat([x, img_feats[i]], dim=1))
            outputs.append(x)
        return outputs

    def forward(self, x, img_feats, defer_last=False):
        """defer_last=True (inference only): returns (outputs of levels 0 .. n-2, finish) where `finish()` -> the last level's
        output.  ds_conv of the last level is already launched; `finish` takes the deepest image-prior level (a stream join when
        the pyramid is still pending) and runs the two blocks that need it.  DepthModel hands `finish` to the decoder, which
        first launches everything that does not depend on it."""
        from . import autograd_ops, ops
        # `img_feats` may be DepthModel's pending pyramid (depth_model._PendingLevels): item i then joins the image-prior encoder's
        # side stream just far enough for level i -- taken AFTER ds_conv_i has been launched, which does not need it
        peek = img_feats.peek() if hasattr(img_feats, "peek") else list(img_feats)
        if autograd_ops.grad_wanted(x, list(peek), self):
            outs = self._forward_train(x, [img_feats[i] for i in range(len(peek))])
            return (outs, None) if defer_last else outs
        outputs = []
        for i in range(self.num_blocks):
            ds = self.convs[f"ds_conv_{i}"]
            c_out = ds.conv2.out_channels
            ho, wo = ops.conv_out_hw(x.shape[2], x.shape[3], ds.stride)
            # concat buffer [x | img_feats[i]] (reference networks.py:124): ds_conv writes its slice in place
            buf = ops.empty_nhwc(x.shape[0], c_out + peek[i].shape[1], ho, wo, x.device)
            ds(x, out=buf[:, :c_out])

            def finish(i=i, buf=buf, c_out=c_out):
                ops.copy_into(buf[:, c_out:], img_feats[i])
                return self.convs[f"conv_{i}"][1](self.convs[f"conv_{i}"][0](buf))
            if defer_last and i == self.num_blocks - 1:
                return outputs, finish
            x = finish()
            outputs.append(x)
        return (outputs, None) if defer_last else outputs


_SIDE_STREAMS = {}  # device -> (stream, stream) for DepthDecoderPP's branch parallelism


class DepthDecoderPP(nn.Module):
    """UNet++ depth decoder (reference networks.py:20-96).  Output heads that the reference evaluates
    and then overwrites (`output_i` is recomputed at every node j, the last one wins, networks.py:92)
    are evaluated once, at the node whose value survives."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([64, 64, 128, 256])
        self.convs = nn.ModuleDict()
        for j in range(1, 5):
            max_i = 4 - j
            for i in range(max_i, -1, -1):
                num_ch_out = int(self.num_ch_dec[i])
                total = 0
                num_ch_in = int(self.num_ch_enc[i + 1] if j == 1 else self.num_ch_dec[i + 1])
                self.convs[f"diag_conv_{i + 1}{j - 1}"] = BasicBlock(num_ch_in, num_ch_out)
                total += num_ch_out
                num_ch_in = int(self.num_ch_enc[i] if j == 1 else self.num_ch_dec[i])
                self.convs[f"right_conv_{i}{j - 1}"] = BasicBlock(num_ch_in, num_ch_out)
                total += num_ch_out
                if i + j != 4:
                    num_ch_in = int(self.num_ch_dec[i + 1])
                    self.convs[f"up_conv_{i + 1}{j}"] = BasicBlock(num_ch_in, num_ch_out)
                    total += num_ch_out
                self.convs[f"in_conv_{i}{j}"] = double_basic_block(total, num_ch_out)
                self.convs[f"output_{i}"] = nn.Sequential(
                    BasicBlock(num_ch_out, num_ch_out) if i != 0 else nn.Identity(),
                    nn.Conv2d(num_ch_out, self.num_output_channels, 1),
                )

    # The three inputs of a node (right / diagonal / up branch) are independent BasicBlocks writing disjoint channel
    # slices of the node's concat buffer.  At small batch a single conv launch leaves CUs idle (150-600 work items for
    # 512 slots), so for batches <= `branch_stream_max_batch` the diagonal and up branches run on two side HIP
    # streams and join before `in_conv`.  At batch 8 the 240x320 launches fill the chip and the fork / join only costs,
    # but the nodes from 120x160 down (<= 1200 of the convs' 8x16-pixel regions for 512 workgroup slots, their diagonal
    # / up branches a quarter of that) leave part of it idle: those fork at any batch (`branch_stream_max_regions`,
    # 0 = off; measured r02 at batch 8: 32.3 -> 32.0 ms per step with 1300, nothing with 400, nothing more with 5000).
    branch_stream_max_batch = 2
    branch_stream_max_regions = int(os.environ.get("SR_DECODER_FORK_REGIONS", "1300"))

    @staticmethod
    def _side_streams(device):
        if device not in _SIDE_STREAMS:  # process-wide, so modules stay picklable / deep-copyable
            _SIDE_STREAMS[device] = (torch.cuda.Stream(device=device), torch.cuda.Stream(device=device))
        return _SIDE_STREAMS[device]

    def _forward_train(self, input_features):
        """Differentiable forward: the UNet++ graph of reference networks.py:75-96 on autograd_ops (torch.cat for the
        concats, the HIP bilinear x2 with its adjoint); the head evaluations the reference overwrites are skipped
        (they receive no gradient there either)."""
        from . import autograd_ops
        prev_outputs, outputs, depth_outputs = list(input_features), [], {}
        for j in range(1, 5):
            for i in range(4 - j, -1, -1):
                parts = [self.convs[f"right_conv_{i}{j - 1}"](prev_outputs[i]),
                         autograd_ops.upsample2x(self.convs[f"diag_conv_{i + 1}{j - 1}"](prev_outputs[i + 1]))]
                if i + j != 4:
                    parts.append(autograd_ops.upsample2x(self.convs[f"up_conv_{i + 1}{j}"](outputs[-1])))
                output = self.convs[f"in_conv_{i}{j}"](torch.cat(parts, dim=1))
                outputs.append(output)
                if i + j == 4:
                    head = self.convs[f"output_{i}"]
                    hx = output if isinstance(head[0], nn.Identity) else head[0](output)
                    depth_outputs[f"log_depth_pred_s{i}_b1hw"] = autograd_ops.conv_bias_act(hx, head[1])
            prev_outputs = outputs[::-1]
        return {k: depth_outputs[k] for k in sorted(depth_outputs, reverse=True)}

    def forward(self, input_features, last_input=None):
        """`last_input` (inference only): a callable returning the deepest input feature, when `input_features` holds all but that
        one (CVEncoder.forward(defer_last=True)).  The first column then launches the right / diagonal branches of the nodes that do
        not read it BEFORE calling it -- the call may join the image-prior encoder's stream, and those branches (the full-resolution
        convolutions among them) run while the encoder finishes.  Same launches, same inputs: the result does not change."""
        from . import autograd_ops, ops
        if last_input is not None and autograd_ops.grad_wanted(list(input_features), self):
            input_features, last_input = list(input_features) + [last_input()], None
        if last_input is None and autograd_ops.grad_wanted(list(input_features), self):
            return self._forward_train(input_features)
        prev_outputs = list(input_features)
        outputs = []
        depth_outputs = {}
        dev = prev_outputs[0].device
        small_batch = dev.type == "cuda" and prev_outputs[0].shape[0] <= self.branch_stream_max_batch
        if dev.type == "cuda":
            main = torch.cuda.current_stream(dev)
            s1, s2 = self._side_streams(dev)
        early = {}   # node i of column 1 -> (buf, fork) whose right / diagonal branches are already launched
        if last_input is not None:
            for i in range(2, -1, -1):
                right = self.convs[f"right_conv_{i}0"]
                c = right.conv2.out_channels
                x_i = prev_outputs[i]
                buf = ops.empty_nhwc(x_i.shape[0], c * 3, x_i.shape[2], x_i.shape[3], x_i.device)
                regions = x_i.shape[0] * ((x_i.shape[2] + 7) // 8) * ((x_i.shape[3] + 15) // 16)
                fork = small_batch or (dev.type == "cuda" and regions <= self.branch_stream_max_regions)
                diag = self.convs[f"diag_conv_{i + 1}0"]
                if fork:
                    s1.wait_stream(main)
                    with torch.cuda.stream(s1):
                        ops.upsample2x(diag(prev_outputs[i + 1]), out=buf[:, c:2 * c])
                    right(x_i, out=buf[:, :c])
                else:
                    right(x_i, out=buf[:, :c])
                    ops.upsample2x(diag(prev_outputs[i + 1]), out=buf[:, c:2 * c])
                early[i] = (buf, fork)
            prev_outputs.append(last_input())
        for j in range(1, 5):
            max_i = 4 - j
            for i in range(max_i, -1, -1):
                right = self.convs[f"right_conv_{i}{j - 1}"]
                diag = self.convs[f"diag_conv_{i + 1}{j - 1}"]
                c = right.conv2.out_channels
                x_i = prev_outputs[i]
                n_parts = 3 if i + j != 4 else 2
                if j == 1 and i in early:
                    # right / diagonal branches are in flight (or done): nothing is left for this stream to run beside the up
                    # branch, so it runs here (no fork / join of a third stream), then the join of the diagonal branch
                    buf, fork = early.pop(i)
                    up = self.convs[f"up_conv_{i + 1}{j}"]
                    ops.upsample2x(up(outputs[-1]), out=buf[:, 2 * c:3 * c])
                    if fork:
                        main.wait_stream(s1)
                    in_conv = self.convs[f"in_conv_{i}{j}"]
                    output = in_conv[1](in_conv[0](buf))
                    outputs.append(output)
                    continue
                buf = ops.empty_nhwc(x_i.shape[0], c * n_parts, x_i.shape[2], x_i.shape[3], x_i.device)
                regions = x_i.shape[0] * ((x_i.shape[2] + 7) // 8) * ((x_i.shape[3] + 15) // 16)
                fork = small_batch or (dev.type == "cuda" and regions <= self.branch_stream_max_regions)
                if fork:
                    s1.wait_stream(main)
                    with torch.cuda.stream(s1):
                        ops.upsample2x(diag(prev_outputs[i + 1]), out=buf[:, c:2 * c])
                    if i + j != 4:
                        s2.wait_stream(main)
                        with torch.cuda.stream(s2):
                            ops.upsample2x(self.convs[f"up_conv_{i + 1}{j}"](outputs[-1]), out=buf[:, 2 * c:3 * c])
                    right(x_i, out=buf[:, :c])
                    main.wait_stream(s1)
                    if i + j != 4:
                        main.wait_stream(s2)
                else:
                    right(x_i, out=buf[:, :c])
                    ops.upsample2x(diag(prev_outputs[i + 1]), out=buf[:, c:2 * c])
                    if i + j != 4:
                        up = self.convs[f"up_conv_{i + 1}{j}"]
                        ops.upsample2x(up(outputs[-1]), out=buf[:, 2 * c:3 * c])
                in_conv = self.convs[f"in_conv_{i}{j}"]
                output = in_conv[1](in_conv[0](buf))
                outputs.append(output)
                if i + j == 4:  # last node of scale i: the head value the reference keeps
                    head = self.convs[f"output_{i}"]
                    hx = output if isinstance(head[0], nn.Identity) else head[0](output)
                    depth_outputs[f"log_depth_pred_s{i}_b1hw"] = ops.conv2d(hx, head[1])
            prev_outputs = outputs[::-1]
        # same key order as the reference's dict (s3, s2, s1, s0 first inserted at j = 1)
        return {k: depth_outputs[k] for k in sorted(depth_outputs, reverse=True)}


class _BlurPool(nn.Module):
    """Parameter/buffer holder for antialiased_cnns.BlurPool(filt_size=4, stride=2) -- keeps the `filt` buffer so
    reference checkpoints load; the HIP kernel uses the fixed outer([1,3,3,1])/64 taps."""

    def __init__(self, channels, filt_size=4, stride=2):
        super().__init__()
        if filt_size != 4 or stride != 2:
            raise ValueError("the HIP path implements BlurPool(filt_size=4, stride=2) only")
        a = torch.tensor([1.0, 3.0, 3.0, 1.0])
        filt = a[:, None] * a[None, :]
        self.register_buffer("filt", (filt / filt.sum())[None, None].repeat(channels, 1, 1, 1))


class _ResnetBlock(nn.Module):
    """Holder for a stride-1 ResNet-18 BasicBlock (conv1, bn1, conv2, bn2)."""

    def __init__(self, planes):
        super().__init__()
        self.conv1 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)


class ResnetMatchingEncoder(nn.Module):
    """Matching-feature encoder (reference networks.py:149-205): antialiased ResNet-18 stem + layer1, then
    conv1x1 64->128, InstanceNorm, LeakyReLU(0.2), conv3x3 128->num_ch_out (replicate padding), InstanceNorm.

    `net` has the reference's nn.Sequential numbering (net.0 conv1, net.1 bn1, net.3.1 blur, net.4 layer1,
    net.5 / net.8 tail convs) so its checkpoints load unchanged.  Inference (eval mode, no gradient wanted): BatchNorm
    uses running statistics folded into the conv weights at pack time, fused kernels.  Training: `_forward_train`, the
    same graph on differentiable operators, BatchNorm per its own mode.  The backbone definition follows the public
    antialiased_cnns package, which is not available here -- parity for it is pinned against a torch.nn
    restatement only (oracle/refshim.py)."""

    def __init__(self, num_layers=18, num_ch_out=16, pretrained=False, antialiased=True):
        super().__init__()
        if num_layers != 18:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers)
                             if num_layers not in (34, 50, 101, 152) else
                             "the HIP path implements the 18-layer backbone (what SimpleRecon uses)")
        if not antialiased:
            raise ValueError("the HIP path implements the antialiased backbone (what SimpleRecon uses)")
        if pretrained:
            raise ValueError("pretrained backbone weights are not downloadable here; load a state_dict instead")
        self.num_ch_enc = np.array([64, 64])
        self.num_ch_out = num_ch_out
        self.net = nn.Sequential(
            nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False),
            nn.BatchNorm2d(64),
            nn.ReLU(inplace=True),
            nn.Sequential(nn.MaxPool2d(kernel_size=2, stride=1), _BlurPool(64)),
            nn.Sequential(_ResnetBlock(64), _ResnetBlock(64)),
            nn.Conv2d(64, 128, (1, 1)),
            nn.InstanceNorm2d(128),
            nn.LeakyReLU(0.2, True),
            nn.Conv2d(128, num_ch_out, (3, 3), padding=1, padding_mode="replicate"),
            nn.InstanceNorm2d(num_ch_out),
        )

    def forward_pair(self, cur_image, src_image):
        """Features of a batch of reference images [B,3,H,W] and their source images [B,K,3,H,W] in ONE pass over
        B(1+K) images without concatenating the images first: the stem writes both groups into one buffer (every
        later layer is per image -- eval BatchNorm, InstanceNorm -- so the order inside the batch is free).
        Returns (cur_feats [B,C,h,w], src_feats [B,K,C,h,w])."""
        from . import ops
        b, k = src_image.shape[:2]
        h, w = cur_image.shape[-2:]
        if self._train_path(cur_image, src_image):
            # training: one batch of B(1+K) images like the reference's TensorFormatter call (depth_model.py:234-240);
            # BatchNorm statistics run over all of them
            feats = self._forward_train(torch.cat([cur_image.unsqueeze(1), src_image], dim=1).flatten(0, 1))
            feats = feats.unflatten(0, (b, 1 + k))
            return feats[:, 0], feats[:, 1:]
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        stem = ops.empty_nhwc(b * (1 + k), 64, ho, wo, cur_image.device)
        ops.stem7x7(cur_image, self.net[0], self.net[1], out=stem[:b])
        ops.stem7x7(src_image.reshape(b * k, *src_image.shape[2:]), self.net[0], self.net[1], out=stem[b:])
        feats = self._after_stem(stem)
        return feats[:b], feats[b:].unflatten(0, (b, k))

    def _train_path(self, *tensors):
        """Training / differentiable path: autograd is recording and something wants a gradient, or a BatchNorm layer is
        in training mode (batch statistics cannot be folded into the conv weights)."""
        from . import autograd_ops
        return autograd_ops.grad_wanted(list(tensors), self) or autograd_ops.any_batchnorm_training(self)

    def _forward_train(self, image):
        """The reference's nn.Sequential (networks.py:176-201) operator by operator on the differentiable HIP operators
        of train_ops: conv1 -> bn1 (batch statistics when the layer is in training mode) + ReLU -> MaxPool(2,1) +
        BlurPool -> layer1 (two BasicBlocks with BatchNorm) -> conv1x1 -> InstanceNorm + LeakyReLU -> replicate-padded
        conv3x3 -> InstanceNorm."""
        from . import train_ops as T
        net = self.net
        x = T.stem7x7(image, net[0])
        x = T.batch_norm_act(x, net[1], act=0.0)
        x = T.maxblurpool(x)
        for blk in net[4]:
            t = T.batch_norm_act(T.conv(x, blk.conv1), blk.bn1, act=0.0)
            t = T.batch_norm_act(T.conv(t, blk.conv2), blk.bn2)
            x = T.add(t, x, act=0.0)
        x = T.conv(x, net[5])
        x = T.instance_norm_act(x, eps=net[6].eps, leaky=net[7].negative_slope)
        x = T.conv(T.replicate_pad(x, 1), net[8], pads=(0, 0, 0, 0))
        return T.instance_norm_act(x, eps=net[9].eps)

    def forward(self, input_image):
        """input_image [B,3,H,W] (H, W multiples of 4) -> [B,num_ch_out,H/4,W/4] (channels_last memory)."""
        from . import ops
        if self._train_path(input_image):
            return self._forward_train(input_image)
        return self._after_stem(ops.stem7x7(input_image, self.net[0], self.net[1]))   # conv1 + bn1 + relu

    def _after_stem(self, x):
        from . import ops
        net = self.net
        x = ops.maxblurpool(x)                                              # MaxPool(2,1) + BlurPool(4,2)
        for blk in net[4]:
            t = ops.conv2d(x, blk.conv1, bn=blk.bn1, leaky=0.0)
            x = ops.conv2d(t, blk.conv2, bn=blk.bn2, residual=x, leaky=0.0)
        fused_in = net[8].out_channels <= 16 and net[8].in_channels % 32 == 0
        if fused_in and (net[5].in_channels, net[5].out_channels) == (64, 128):
            # the 1x1 conv leaves the InstanceNorm statistics of its output behind (no separate pass over it) ...
            x, stats = ops.conv1x1_stats(x, net[5], eps=net[6].eps)
            x = ops.conv3x3_c16(x, net[8], in_stats=stats, in_leaky=net[7].negative_slope)
        elif fused_in:
            # ... and InstanceNorm + LeakyReLU of the 128-channel map are applied inside the last conv's input
            # staging: the normalised tensor never goes to HBM
            x = ops.conv2d(x, net[5])
            stats = ops.instance_norm_stats(x, eps=net[6].eps)
            x = ops.conv3x3_c16(x, net[8], in_stats=stats, in_leaky=net[7].negative_slope)
        else:
            x = ops.conv2d(x, net[5])
            x = ops.instance_norm(x, eps=net[6].eps, leaky=net[7].negative_slope, inplace=True)
            x = ops.conv2d(x, net[8])
        return ops.instance_norm(x, eps=net[9].eps, inplace=True)
