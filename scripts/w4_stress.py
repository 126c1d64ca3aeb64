"""Stress of the wave-specialised F(4x4) kernel next to other kernels on side streams (co-resident work skews the waves of a
workgroup): every output must equal the quiet 4-wave form bit for bit, many repetitions, shapes with 1 .. 12 slabs, borders, ragged
sizes, with and without residual / bias, LeakyReLU / none / SiLU.    python scripts/w4_stress.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_wino4 import _run
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
shapes = [(2, 16, 64, 80, 64, True, 0.2), (3, 16, 33, 47, 64, False, 0.2), (2, 64, 120, 160, 64, True, 0.2), (2, 128, 60, 80, 128, True, 0.2),
          (1, 64, 240, 320, 64, True, 0.2), (1, 192, 48, 64, 64, False, None), (4, 24, 35, 53, 72, True, 0.0), (2, 48, 16, 16, 64, False, 0.2)]
torch.manual_seed(11)
cases = []
for (b, ci, h, w, co, with_res, leaky) in shapes:
    conv = torch.nn.Conv2d(ci, co, 3, padding=1).to(DEV)
    x = torch.randn(b, ci, h, w, device=DEV).contiguous(memory_format=torch.channels_last)
    res = torch.randn(b, co, h, w, device=DEV).contiguous(memory_format=torch.channels_last) if with_res else None
    with torch.inference_mode():
        want = _run("w4", x, conv, res, leaky).clone()
        _run("w2", x, conv, res, leaky)
    cases.append((conv, x, res, leaky, want))
noise = torch.randn(1 << 22, device=DEV)
torch.cuda.synchronize()
main, s1, s2 = (torch.cuda.Stream(device=DEV) for _ in range(3))
bad = 0
with torch.inference_mode():
    for rep in range(reps):
        outs = []
        for i, (conv, x, res, leaky, _) in enumerate(cases):
            with torch.cuda.stream(s1):
                for _ in range(3):
                    noise = torch.sin(noise) * 1.0001
            with torch.cuda.stream(s2):
                j = (i + 1) % len(cases)
                _run("w2", cases[j][1], cases[j][0], None, cases[j][3])
            with torch.cuda.stream(main):
                outs.append(_run("w4_ws", x, conv, res, leaky))
        torch.cuda.synchronize()
        for i, got in enumerate(outs):
            if not torch.equal(got, cases[i][4]):
                bad += 1
                print(f"rep {rep} shape {shapes[i]}: {int((got != cases[i][4]).sum())} elements differ", flush=True)
print(f"{reps} repetitions x {len(cases)} shapes: {bad} mismatching launches")
sys.exit(1 if bad else 0)
