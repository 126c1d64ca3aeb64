import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_graph import _model, _inputs
from simplerecon_amd import networks
B, K, D, h, w = 4, 3, 8, 24, 32
a = _inputs(B, K, h, w, 5)

def run(model, lo, hi):
    sl = lambda t: [f[lo:hi] for f in t] if isinstance(t, list) else t[lo:hi]
    mc, ms = model.compute_matching_feats(a[0], a[1], False)
    return model.hot_path(sl(a[2]), mc[lo:hi], ms[lo:hi], *[sl(t) for t in a[3:]], return_mask=True)

for fork_max in (2, 0):
    networks.DepthDecoderPP.branch_stream_max_batch = fork_max
    for streams in (2, 4):
        bounds = [(B * i) // streams for i in range(streams + 1)]
        seq_model = _model(h, w, K, D)
        model = _model(h, w, K, D)
        with torch.inference_mode():
            parts = [run(seq_model, bounds[i], bounds[i + 1]) for i in range(streams)]
            want = {k: torch.cat([p[k] for p in parts], 0) for k in parts[0] if parts[0][k] is not None}
            # determinism of the sequential path itself
            parts2 = [run(seq_model, bounds[i], bounds[i + 1]) for i in range(streams)]
            want2 = {k: torch.cat([p[k] for p in parts2], 0) for k in parts2[0] if parts2[0][k] is not None}
            model.num_streams = streams
            for rep in range(3):
                mc, ms = model.compute_matching_feats(a[0], a[1], False)
                out = model.hot_path(list(a[2]), mc, ms, *a[3:], return_mask=True)
                torch.cuda.synchronize()
                bad = {k: float((out[k].float() - want[k].float()).abs().max()) for k in want if not torch.equal(out[k], want[k])}
                print(f"fork_max {fork_max} streams {streams} rep {rep}: mismatching keys {bad}", flush=True)
        print("  sequential path deterministic:", all(torch.equal(want[k], want2[k]) for k in want), flush=True)
