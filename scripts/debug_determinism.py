"""Run-to-run identity of DepthModel.hot_path with the caching allocator full of poisoned blocks (a read of memory no kernel
of the run wrote shows up as a difference / NaN), module by module: prints the first modules whose output differs."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from simplerecon_amd import depth_model as dm, synthetic

DEV = "cuda:0"
POISON = os.environ.get("SR_DBG_POISON", "rand")


def poison(total_gb=24):
    blocks, left = [], int(total_gb * 2 ** 30)
    g = torch.Generator(device=DEV).manual_seed(int(os.environ.get("SR_DBG_SEED", "1")))
    sizes = [2 ** k for k in range(12, 30)]
    i = 0
    while left > 0:
        n = sizes[i % len(sizes)] // 4
        t = torch.empty(n, device=DEV)
        if POISON == "nan":
            t.fill_(float("nan"))
        else:
            t.uniform_(-3.0, 3.0, generator=g)
        blocks.append(t)
        left -= 4 * n
        i += 1
    torch.cuda.synchronize()
    del blocks


def main():
    B, K, C, D, h, w = int(os.environ.get("SR_DBG_B", 2)), 7, 16, 64, 120, 160
    opts = dm.default_options(image_width=4 * w, image_height=4 * h, model_num_views=K + 1, matching_num_depth_bins=D)
    model = dm.DepthModel(opts)
    synthetic.seeded_fill_(model.cost_volume_net, seed=1)
    synthetic.seeded_fill_(model.depth_decoder, seed=2)
    synthetic.seeded_fill_(model.cost_volume.mlp, seed=3)
    model = model.to(DEV).eval()
    inp = synthetic.cost_volume_inputs(B, K, C, h, w, seed=11, device=DEV)
    pyr = synthetic.image_prior_pyramid(B, h, w, seed=11, device=DEV)
    cur = []

    def hook(name):
        def f(mod, args, out):
            if torch.is_tensor(out):
                cur.append((name, out.detach().clone()))
        return f
    for name, m in model.named_modules():
        if name and (not list(m.children()) or type(m).__name__ == "BasicBlock"):
            m.register_forward_hook(hook(name))

    def run():
        cur.clear()
        with torch.inference_mode():
            out = model.hot_path(list(pyr), inp["cur_feats"], inp["src_feats"], inp["src_extrinsics"], inp["src_poses"],
                                 inp["src_Ks"], inp["cur_invK"], return_mask=True)
        torch.cuda.synchronize()
        return out, list(cur)
    runs = []
    for r in range(int(os.environ.get("SR_DBG_RUNS", 3))):
        if POISON != "none":
            poison()
        runs.append(run())
    print("hooked tensors per run:", [len(c) for _, c in runs])
    for r in range(1, len(runs)):
        bad = 0
        for (na, ta), (nb, tb) in zip(runs[0][1], runs[r][1]):
            assert na == nb
            if not torch.equal(ta, tb):
                d = (ta - tb).abs()
                nn_ = int((ta != tb).sum())
                print(f"run {r}: {na:40s} {tuple(ta.shape)} differs in {nn_} elements, max |d| "
                      f"{float(d.nan_to_num(1e30).max()):.3e} nan {int(torch.isnan(tb).sum())}/{int(torch.isnan(ta).sum())}")
                print("     first:", (ta != tb).nonzero()[:4].tolist())
                bad += 1
                if bad >= 6:
                    break
        for k in runs[0][0]:
            if torch.is_tensor(runs[0][0][k]) and not torch.equal(runs[0][0][k], runs[r][0][k]):
                print(f"run {r}: output {k} differs")
        print(f"run {r}: {bad} differing modules (first 6 shown)")


main()
