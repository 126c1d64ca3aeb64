"""Times the stages of the native matching encoder (a16) on the GPU: python scripts/menc_micro.py [n_images]"""
import sys

import torch

sys.path.insert(0, ".")
from simplerecon_amd import ops, synthetic  # noqa: E402
from simplerecon_amd.networks import ResnetMatchingEncoder  # noqa: E402

DEV = "cuda:0"


def timed(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    nimg = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    enc = synthetic.seeded_fill_(ResnetMatchingEncoder(18, 16), seed=1).to(DEV).eval()
    x = torch.randn((nimg, 3, 480, 640), device=DEV)
    net = enc.net
    with torch.inference_mode():
        stem = ops.stem7x7(x, net[0], net[1])
        pool = ops.maxblurpool(stem)
        blk = net[4][0]
        t = ops.conv2d(pool, blk.conv1, bn=blk.bn1, leaky=0.0)
        c1 = ops.conv2d(pool, net[5])
        c2 = ops.conv2d(c1, net[8])
        st = ops.instance_norm_stats(c1)
        rows = [
            ("stem 7x7 s2 + bn + relu", lambda: ops.stem7x7(x, net[0], net[1]), 2.0 * nimg * 240 * 320 * 64 * 147),
            ("maxpool + blurpool", lambda: ops.maxblurpool(stem), None),
            ("layer1 conv3x3 64->64 (x4)", lambda: ops.conv2d(pool, blk.conv1, bn=blk.bn1, leaky=0.0),
             2.0 * nimg * 120 * 160 * 64 * 64 * 9),
            ("layer1 conv3x3 + residual", lambda: ops.conv2d(t, blk.conv2, bn=blk.bn2, residual=pool, leaky=0.0),
             2.0 * nimg * 120 * 160 * 64 * 64 * 9),
            ("conv1x1 64->128", lambda: ops.conv2d(pool, net[5]), 2.0 * nimg * 120 * 160 * 64 * 128),
            ("instance norm 128 + lrelu", lambda: ops.instance_norm(c1, leaky=0.2, inplace=True), None),
            ("conv3x3 128->16 replicate", lambda: ops.conv2d(c1, net[8]), 2.0 * nimg * 120 * 160 * 128 * 16 * 9),
            ("instance norm stats 128", lambda: ops.instance_norm_stats(c1), None),
            ("norm+lrelu+conv3x3 128->16 fused", lambda: ops.conv3x3_c16(c1, net[8], in_stats=st, in_leaky=0.2),
             2.0 * nimg * 120 * 160 * 128 * 16 * 9),
            ("instance norm 16", lambda: ops.instance_norm(c2, inplace=True), None),
            ("whole encoder", lambda: enc(x), 8.1e9 * nimg),
        ]
        for name, fn, flops in rows:
            ms = timed(fn)
            extra = f"  {flops / ms / 1e9:7.1f} TF" if flops else ""
            print(f"{name:32s} {ms:8.3f} ms{extra}")


if __name__ == "__main__":
    main()
