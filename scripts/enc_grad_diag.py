import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import numpy as np, torch
import test_gpu_encoder_training as t
from simplerecon_amd import image_encoder, synthetic, networks
import effnet_torch, golden_cases as gc
DEV = "cuda:0"
def rms(x): 
    x = x.detach().cpu().double().numpy() if isinstance(x, torch.Tensor) else np.asarray(x, np.float64)
    return float(np.sqrt((x**2).mean()))
# matching encoder
for mode in ("train", "eval"):
    case = gc.MATCHING_CASES["small"]
    gold = np.load(os.path.join(t.GOLDEN, f"grad_matching_encoder_{mode}.npz"))
    enc = networks.ResnetMatchingEncoder(18, 16); synthetic.seeded_fill_(enc, seed=case["seed"]); enc = enc.to(DEV); enc.train(mode == "train")
    x = gc.matching_input(case).to(DEV); y = enc(x)
    cot = torch.from_numpy(gc.encoder_cotangent(case, tuple(y.shape))).to(DEV); (y * cot).sum().backward()
    print(mode, "fwd rel", t.rel_err(y, gold["out"]))
    for n, p in enc.named_parameters():
        r = gold["d_" + n]
        print(f"  {n:28s} ref rms {rms(r):.3e} ours rms {rms(p.grad):.3e} err rms {rms(p.grad.cpu().numpy() - r):.3e} relL2 {t.rel_l2(p.grad, r):.3e}")
