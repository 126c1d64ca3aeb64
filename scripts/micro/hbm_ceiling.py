"""What a plain stream reaches on this box (HIP events): ATen copy / fill / sum on 1.26 GB, the size of the matching encoder's
[64, 64, 240, 320] map -- the practical ceiling the HBM-bound kernels of profiles/README.md are priced against."""
import torch
dev = "cuda:0"
n = 64 * 64 * 240 * 320
a = torch.randn(n, device=dev)
b = torch.empty_like(a)


def t(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


gb = n * 4 / 1e9
for name, f, bytes_ in (("copy (read + write)", lambda: b.copy_(a), 2 * gb), ("fill (write)", lambda: b.fill_(1.0), gb),
                        ("sum (read)", lambda: a.sum(), gb), ("a*2 -> b", lambda: torch.mul(a, 2.0, out=b), 2 * gb),
                        ("copy 4:1 (read 4, write 1)", lambda: b[: n // 4].copy_(a.view(4, -1).sum(0)) if False else torch.sum(a.view(4, -1), 0, out=b[: n // 4]), 1.25 * gb)):
    s = t(f)
    print(f"{name:28s} {s*1e6:8.1f} us  {bytes_/s/1e3:6.2f} TB/s")
