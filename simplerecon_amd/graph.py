"""HIP-graph replay of a launch-bound step.

One keyframe batch is ~230 kernel launches (matching encoder + plane sweep + 165 convs); at batch 1 the Python /
ctypes submission of those launches (~6 ms) is as long as the GPU work.  `GraphedCallable` records the launches of
`fn(*inputs)` once into a HIP graph (`torch.cuda.CUDAGraph` is hipGraph on ROCm) and replays it: one submission per
step, no per-launch host work, back-to-back kernels on the device.

Everything the HIP path launches goes to torch's current stream (`_lib.stream_ptr`), so capture needs no special
casing; the library itself never allocates or synchronises.  Packed-weight caches and workspaces are filled by the
warm-up runs before the capture, inside it they are only read.
"""
import torch


def _flatten(x, out):
    if isinstance(x, torch.Tensor):
        out.append(x)
    elif isinstance(x, (list, tuple)):
        for v in x:
            _flatten(v, out)
    elif isinstance(x, dict):
        for k in x:
            _flatten(x[k], out)
    elif x is not None and not isinstance(x, (bool, int, float, str)):
        raise TypeError(f"GraphedCallable inputs must be tensors / lists / dicts / scalars, got {type(x)}")
    return out


def _clone_like(x):
    if isinstance(x, torch.Tensor):
        return x.clone(memory_format=torch.preserve_format)
    if isinstance(x, (list, tuple)):
        return type(x)(_clone_like(v) for v in x)
    if isinstance(x, dict):
        return {k: _clone_like(v) for k, v in x.items()}
    return x


class GraphedCallable:
    """`g = GraphedCallable(fn, *example_inputs)`; `out = g(*inputs)` copies `inputs` into the captured input
    buffers (device-to-device, skipped for tensors that already ARE those buffers: see `static_inputs`) and replays the
    graph.  The returned tensors are the graph's static output buffers: they are overwritten by the next call.
    Shapes, dtypes, devices and non-tensor arguments are frozen at capture time."""

    def __init__(self, fn, *example_inputs, warmup=3):
        flat = _flatten(example_inputs, [])
        if not flat or not all(t.is_cuda for t in flat):
            raise ValueError("GraphedCallable needs device tensors as inputs")
        self.device = flat[0].device
        self.static_inputs = _clone_like(example_inputs)
        self._static_flat = _flatten(self.static_inputs, [])
        self._shapes = [(tuple(t.shape), t.dtype) for t in self._static_flat]
        with torch.cuda.device(self.device), torch.inference_mode():
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):   # fills weight-pack caches and workspaces outside the capture
                    fn(*self.static_inputs)
            torch.cuda.current_stream(self.device).wait_stream(side)
            torch.cuda.synchronize(self.device)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_outputs = fn(*self.static_inputs)

    def __call__(self, *inputs):
        flat = _flatten(inputs, [])
        if len(flat) != len(self._static_flat):
            raise ValueError(f"expected {len(self._static_flat)} input tensors, got {len(flat)}")
        with torch.inference_mode():
            for dst, src, (shape, dtype) in zip(self._static_flat, flat, self._shapes):
                if src is dst:
                    continue
                if tuple(src.shape) != shape or src.dtype != dtype:
                    raise ValueError(f"input of shape {tuple(src.shape)} / {src.dtype} does not match the captured "
                                     f"{shape} / {dtype}")
                dst.copy_(src, non_blocking=True)
            self.graph.replay()
        return self.static_outputs
