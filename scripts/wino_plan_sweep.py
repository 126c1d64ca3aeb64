"""Launch-plan sweep of the Winograd conv: every 3x3 shape of the hero model's conv stack at batch 8 and 1, timed under
each forced (NT, split-K) plan in its own process (the env switches are read once) next to the default plan's choice."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 240, 320, 64), (192, 240, 320, 64), (128, 240, 320, 64), (24, 240, 320, 64), (64, 120, 160, 64),
          (192, 120, 160, 64), (128, 120, 160, 64), (112, 120, 160, 64), (128, 60, 80, 128), (128, 60, 80, 64),
          (64, 60, 80, 64), (384, 60, 80, 128), (256, 60, 80, 128), (192, 60, 80, 128), (256, 30, 40, 256),
          (256, 30, 40, 128), (128, 30, 40, 128), (512, 30, 40, 256), (416, 30, 40, 256), (384, 15, 20, 384),
          (384, 15, 20, 256), (640, 15, 20, 384), (256, 15, 20, 256)]
PLANS = [(0, 0), (2, 1), (1, 1), (2, 2), (1, 2), (2, 4), (1, 4), (2, 8), (1, 8)]
if os.environ.get("SR_SWEEP_PLANS"):   # e.g. "0:0,2:1,1:1"
    PLANS = [tuple(int(v) for v in t.split(":")) for t in os.environ["SR_SWEEP_PLANS"].split(",")]


def child():
    sys.path.insert(0, ROOT)
    import torch
    from simplerecon_amd import ops
    dev = "cuda:0"
    for B in (8, 1):
        for (ci, H, W, co) in SHAPES:
            conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(dev)
            x = torch.randn(B, ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
            out = ops.empty_nhwc(B, co, H, W, dev)
            with torch.inference_mode():
                f = lambda: ops.conv2d(x, conv, leaky=0.2, out=out)
                for _ in range(3): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 30
                e0.record()
                for _ in range(n): f()
                e1.record(); e1.synchronize()
            print(f"R {B} {ci} {H} {W} {co} {e0.elapsed_time(e1) * 1e3 / n:.2f}", flush=True)


if __name__ == "__main__":
    if os.environ.get("SR_SWEEP_CHILD"):
        child()
        sys.exit(0)
    table = {}
    for nt, ks in PLANS:
        env = dict(os.environ, SR_SWEEP_CHILD="1", SR_CONV_WINO="2")
        if nt: env["SR_WINO_NT"] = str(nt)
        if ks: env["SR_WINO_KSPLIT"] = str(ks)
        r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("R "):
                v = line.split()
                table.setdefault(tuple(int(a) for a in v[1:6]), {})[(nt, ks)] = float(v[6])
        if r.returncode: print(r.stderr[-2000:])
    print("(B,Ci,H,W,Co)".ljust(28) + "".join(f"{('default' if not nt else f'nt{nt}ks{ks}'):>9s}" for nt, ks in PLANS) + "   best  gain")
    tot_d = tot_b = 0.0
    for k, row in table.items():
        d = row.get((0, 0), float("nan"))
        best = min(row, key=row.get)
        print(str(k).ljust(28) + "".join(f"{row.get(pl, float('nan')):9.1f}" for pl in PLANS) +
              f"   nt{best[0]}ks{best[1]} {100 * (d - row[best]) / d:5.1f}%")
