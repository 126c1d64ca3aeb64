"""Tuning sweep of the LDS-staged dot-product sweep (csrc/sr_dot_volume_lds.hip): one child process per variant
(the switches are read once per process), each timing sr_dot_volume_sweep at cfg2 shapes and comparing its volume /
lowest cost / mask with the L1-gather kernel (SR_DOT_LDS=0).

    python scripts/dot_lds_sweep.py            # parent: runs every variant
"""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    import torch
    from simplerecon_amd import _lib, synthetic
    from simplerecon_amd.cost_volume import CostVolumeManager
    dev = torch.device("cuda", 0)
    out = {"env": {k: v for k, v in os.environ.items() if k.startswith("SR_DOT")}}
    lib = _lib.lib()
    for B in (1, 8):
        K, Cc, h, w, D = 7, 16, 120, 160, 64
        inp = synthetic.cost_volume_inputs(B, K, Cc, h, w, seed=0, device=dev)
        m = CostVolumeManager(h, w, num_depth_bins=D).to(dev)
        planes = m.generate_depth_planes(B, inp["min_depth"], inp["max_depth"])
        vol = torch.zeros((B, D, h, w), device=dev)
        lowest = torch.zeros((B, h, w), device=dev)
        mask = torch.zeros((B, h, w), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.sr_volume_workspace_bytes(B, K, Cc, h, w), dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr(dev)
        _lib.check(lib.sr_volume_prepare(_lib.ptr(inp["src_feats"]), _lib.ptr(inp["src_Ks"]),
                                         _lib.ptr(inp["src_extrinsics"]), None, B, K, Cc, h, w, _lib.ptr(ws),
                                         ws.numel(), st), "prepare")

        def sweep():
            _lib.check(lib.sr_dot_volume_sweep(_lib.ptr(inp["cur_feats"]), _lib.ptr(inp["cur_invK"]), _lib.ptr(planes),
                                               *planes.stride(), B, K, Cc, h, w, D, _lib.ptr(vol), D * h * w, h * w, 1,
                                               _lib.ptr(lowest), _lib.ptr(mask), _lib.ptr(ws), ws.numel(), st), "sweep")
        for _ in range(3):
            sweep()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            sweep()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        out[f"B{B}_us_per_call"] = us
        out[f"B{B}_us_per_frame"] = us / B
        ref = os.path.join("/tmp", f"dot_ref_B{B}.npz")
        v, lo, mk = vol.cpu().numpy(), lowest.cpu().numpy(), mask.cpu().numpy()
        if os.environ.get("SR_DOT_LDS") == "0":
            np.savez(ref, v=v, lo=lo, mk=mk)
        elif os.path.exists(ref):
            r = np.load(ref)
            out[f"B{B}_max_abs_diff"] = float(np.abs(v - r["v"]).max())
            out[f"B{B}_vol_absmax"] = float(np.abs(r["v"]).max())
            out[f"B{B}_lowest_mismatch"] = int((lo != r["lo"]).sum())
            out[f"B{B}_mask_mismatch"] = int((mk != r["mk"]).sum())
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if os.environ.get("SR_SWEEP_CHILD"):
        return child()
    variants = [{"SR_DOT_LDS": "0"}]
    for cap, g in (("634", "4"), ("634", "2"), ("506", "2"), ("506", "4")):
        variants.append({"SR_DOT_LDS": "1", "SR_DOT_LDS_CAP": cap, "SR_DOT_LDS_G": g})
    for v in variants:
        env = dict(os.environ, SR_SWEEP_CHILD="1", **v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True,
                           timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        print(line[0] if line else f"FAILED {v}: rc={r.returncode} {r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    main()
