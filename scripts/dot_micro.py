"""A few launches of the dot-product sweep at cfg2 shape (batch SR_MICRO_B, default 8) -- target of rocprofv3 passes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simplerecon_amd import _lib, synthetic
from simplerecon_amd.cost_volume import CostVolumeManager

B = int(os.environ.get("SR_MICRO_B", "8"))
n = int(os.environ.get("SR_MICRO_N", "5"))
K, Cc, h, w, D = 7, 16, 120, 160, 64
dev = torch.device("cuda", 0)
lib = _lib.lib()
inp = synthetic.cost_volume_inputs(B, K, Cc, h, w, seed=0, device=dev)
m = CostVolumeManager(h, w, num_depth_bins=D).to(dev)
planes = m.generate_depth_planes(B, inp["min_depth"], inp["max_depth"])
vol = torch.zeros((B, D, h, w), device=dev)
lowest = torch.zeros((B, h, w), device=dev)
ws = torch.empty(lib.sr_volume_workspace_bytes(B, K, Cc, h, w), dtype=torch.uint8, device=dev)
st = _lib.stream_ptr(dev)
_lib.check(lib.sr_volume_prepare(_lib.ptr(inp["src_feats"]), _lib.ptr(inp["src_Ks"]), _lib.ptr(inp["src_extrinsics"]), None,
                                 B, K, Cc, h, w, _lib.ptr(ws), ws.numel(), st), "prepare")
for _ in range(n):
    _lib.check(lib.sr_dot_volume_sweep(_lib.ptr(inp["cur_feats"]), _lib.ptr(inp["cur_invK"]), _lib.ptr(planes),
                                       *planes.stride(), B, K, Cc, h, w, D, _lib.ptr(vol), D * h * w, h * w, 1,
                                       _lib.ptr(lowest), None, _lib.ptr(ws), ws.numel(), st), "sweep")
torch.cuda.synchronize()
