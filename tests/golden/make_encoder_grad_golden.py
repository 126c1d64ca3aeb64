"""Generates tests/golden/grad_matching_encoder_{train,eval}.npz: forward output, every parameter gradient and (training
mode) the updated BatchNorm running statistics of the REFERENCE's ResnetMatchingEncoder (modules/networks.py:149-205, on
oracle/refshim.py's torch.nn restatement of the absent antialiased_cnns backbone), from the reference's own autograd on
CPU in float64 (build container only; needs /root/reference):

    python tests/golden/make_encoder_grad_golden.py

loss = sum(features * R), R seeded (tests/golden_cases.py::encoder_cotangent).  float64: with ReLU / max-pool decisions
in the graph an fp32 run differs from any other implementation by the handful of elements that sit on a kink."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import refshim  # noqa: E402
import golden_cases as gc  # noqa: E402
from simplerecon_amd import synthetic  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    torch.set_num_threads(8)
    nets = refshim.import_reference()[1]
    case = gc.MATCHING_CASES["small"]
    for mode in ("train", "eval"):
        enc = nets.ResnetMatchingEncoder(18, 16, pretrained=False)
        synthetic.seeded_fill_(enc, seed=case["seed"])
        enc = enc.double()
        enc.train(mode == "train")
        x = gc.matching_input(case).double()
        y = enc(x)
        cot = torch.from_numpy(gc.encoder_cotangent(case, tuple(y.shape))).double()
        (y * cot).sum().backward()
        save = {"out": y.detach().numpy().astype(np.float32)}
        save.update({"d_" + k: p.grad.numpy().astype(np.float32) for k, p in enc.named_parameters()})
        if mode == "train":
            save.update({"buf_" + k: b.detach().numpy().astype(np.float32) for k, b in enc.named_buffers() if "running" in k})
        np.savez_compressed(os.path.join(OUT, f"grad_matching_encoder_{mode}.npz"), **save)
        print(mode, tuple(y.shape), {k: v.shape for k, v in list(save.items())[:4]}, len(save))


if __name__ == "__main__":
    main()
