"""Who launches the non-sr_* kernels of a step?  (~100 __amd_rocclr_copyBuffer + ATen launches per step.)

    python scripts/attrib_launches.py [workload] > gpurun_out/attrib.txt

Every ATen operator the step dispatches (a TorchDispatchMode sees them all: copy_, clone, cat, exp ...) is counted per
(operator, innermost frame of this repository on the Python stack), after warm-up.  The C-ABI kernels do not go through the
dispatcher, so what is listed here is exactly the torch-side work of a step: device copies (`__amd_rocclr_copyBuffer` is
what a contiguous device-to-device `copy_` / `clone` / `cat` becomes) and elementwise kernels."""
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench_workloads  # noqa: E402

NO_KERNEL = ("aten::view", "aten::as_strided", "aten::slice", "aten::select", "aten::expand", "aten::permute", "aten::t",
             "aten::transpose", "aten::unsqueeze", "aten::squeeze", "aten::detach", "aten::alias", "aten::_unsafe_view",
             "aten::empty", "aten::empty_like", "aten::empty_strided", "aten::split", "aten::unbind", "aten::reshape",
             "aten::flatten", "aten::narrow", "aten::unflatten", "aten::_reshape_alias", "aten::new_empty", "aten::size",
             "aten::stride", "aten::is_contiguous", "aten::sym_size", "aten::lift_fresh", "aten::record_stream")


class Count(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.n = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func._schema.name
        if not name.startswith(NO_KERNEL):
            frames = [f for f in traceback.extract_stack() if ("simplerecon_amd" in f.filename or "bench_workloads" in f.filename)]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in reversed(frames[-3:])) or "?"
            shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            self.n[(name, where, shape)] += 1
        return func(*args, **(kwargs or {}))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else bench_workloads.DEFAULT
    steps = 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    wl = bench_workloads.WORKLOADS[name](dev, 0)
    with torch.inference_mode():
        for _ in range(3):
            wl.step()
        torch.cuda.synchronize()
        with Count() as c:
            for i in range(steps):
                wl.step(i)
        torch.cuda.synchronize()
    print(f"workload {name}: ATen operators dispatched per step (over {steps} steps; views / allocations not listed)")
    tot = 0
    for key, n in sorted(c.n.items(), key=lambda kv: -kv[1]):
        tot += n
        print(f"{n / steps:7.1f}/step  {key[0]:24s} {str(key[2]):28s} {key[1]}")
    print(f"total {tot / steps:.1f} per step")


if __name__ == "__main__":
    main()
