"""Times the image-prior encoder (EfficientNetV2-S pyramid) per stage on the GPU: python scripts/effnet_micro.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simplerecon_amd import ops, synthetic
from simplerecon_amd.image_encoder import EfficientNetV2SFeatures

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda:0"
enc = synthetic.seeded_fill_(EfficientNetV2SFeatures(), seed=3, gain=1.0).to(dev)
img = torch.randn(B, 3, 480, 640, device=dev)
FLOPS = 0.0


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

with torch.inference_mode():
    x = ops.rgb_stem3x3s2(img, enc.conv_stem, bn=enc.bn1, act="silu", tf_same=True)
    print(f"stem   {timed(lambda: ops.rgb_stem3x3s2(img, enc.conv_stem, bn=enc.bn1, act='silu', tf_same=True)):8.3f} ms")
    tot = 0.0
    for i, stage in enumerate(enc.blocks):
        xin = x
        ops.PROFILE = []
        x = stage(xin)
        torch.cuda.synchronize()
        fl = sum(r[1] for r in ops.PROFILE)
        ops.PROFILE = None
        t = timed(lambda: stage(xin))
        tot += t
        print(f"stage{i} {t:8.3f} ms  {fl/1e9:7.1f} GFLOP  {fl/t/1e9:6.1f} TF  out {tuple(x.shape)}")
        if i >= 3:
            blk = stage[1]
            y = stage[0](xin)
            t1 = timed(lambda: ops.conv2d(y, blk.conv_pw, bn=blk.bn1, act="silu"))
            tt = ops.conv2d(y, blk.conv_pw, bn=blk.bn1, act="silu")
            t2 = timed(lambda: ops.dwconv3x3(tt, blk.conv_dw, bn=blk.bn2, act="silu", tf_same=True, want_pool=True))
            d, pool = ops.dwconv3x3(tt, blk.conv_dw, bn=blk.bn2, act="silu", tf_same=True, want_pool=True)
            t3 = timed(lambda: ops.se_gate(pool, d.shape[2] * d.shape[3], blk.se.conv_reduce, blk.se.conv_expand))
            g = ops.se_gate(pool, d.shape[2] * d.shape[3], blk.se.conv_reduce, blk.se.conv_expand)
            t4 = timed(lambda: ops.scale_channels_(d, g))
            t5 = timed(lambda: ops.conv2d(d, blk.conv_pwl, bn=blk.bn3, residual=y))
            print(f"   block1: pw {t1*1e3:.0f} us  dw {t2*1e3:.0f} us  se {t3*1e3:.0f} us  scale {t4*1e3:.0f} us  pwl {t5*1e3:.0f} us")
            if ops.mbconv_fused_supported(y, blk.conv_pw, blk.conv_dw, blk.se):
                t6 = timed(lambda: ops.mbconv_expand_dw_se(y, blk.conv_pw, blk.bn1, blk.conv_dw, blk.bn2, blk.se))
                dd, gg = ops.mbconv_expand_dw_se(y, blk.conv_pw, blk.bn1, blk.conv_dw, blk.bn2, blk.se)
                t7 = timed(lambda: ops.conv2d(dd, blk.conv_pwl, bn=blk.bn3, residual=y, gate=gg))
                print(f"   block1 fused: expansion+depthwise+gates {t6*1e3:.0f} us  gated projection {t7*1e3:.0f} us")
    print(f"whole encoder {timed(lambda: enc(img)):8.3f} ms for {B} images (stages sum {tot:.3f})")
