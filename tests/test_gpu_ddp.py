"""Data-parallel training of the whole DepthModel (reference train.py:126-142 trains under DDP): two ranks on the one GPU
of the test box (gloo: RCCL refuses two ranks on one device; on a node the same code runs with backend "nccl", one rank
per GPU), torch.nn.parallel.DistributedDataParallel around simplerecon_amd.DepthModel with BatchNorm in training mode.
Each rank sees different keyframes; after backward every rank must hold the AVERAGE of the two ranks' local gradients --
checked against the gradients each rank computes on its own without DDP."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batch(rank, B, K, H, W, dev):
    import numpy as np
    from simplerecon_amd import synthetic
    inp = synthetic.cost_volume_inputs(B, K, 16, H // 4, W // 4, seed=10 + rank, device=dev)
    rng = np.random.default_rng(100 + rank)
    eye = torch.eye(4, device=dev).expand(B, 4, 4).contiguous()
    cur = {"image_b3hw": torch.from_numpy(rng.standard_normal((B, 3, H, W)).astype("float32")).to(dev),
           "invK_s1_b44": inp["cur_invK"], "cam_T_world_b44": eye, "world_T_cam_b44": eye}
    src = {"image_b3hw": torch.from_numpy(rng.standard_normal((B, K, 3, H, W)).astype("float32")).to(dev),
           "K_s1_b44": inp["src_Ks"], "cam_T_world_b44": inp["src_extrinsics"], "world_T_cam_b44": inp["src_poses"]}
    return cur, src


def _loss(out):
    return sum(out[f"log_depth_pred_s{i}_b1hw"].abs().mean() for i in range(4))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from simplerecon_amd import depth_model as dm
        from simplerecon_amd import synthetic
        dev = torch.device("cuda", 0)
        B, K, H, W, D = 1, 2, 64, 96, 8
        opts = dm.default_options(image_width=W, image_height=H, model_num_views=K + 1, matching_num_depth_bins=D)
        model = dm.DepthModel(opts)
        synthetic.seeded_fill_(model.encoder, seed=6, gain=1.0)
        for i, m in enumerate((model.matching_model, model.cost_volume_net, model.depth_decoder, model.cost_volume.mlp)):
            synthetic.seeded_fill_(m, seed=20 + i)
        model = model.to(dev).train()
        cur, src = _batch(rank, B, K, H, W, dev)
        # local gradients, no DDP
        torch.manual_seed(500 + rank)
        _loss(model("train", cur, src)).backward()
        local = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        model.zero_grad(set_to_none=True)
        # the same step under DDP
        ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], find_unused_parameters=True)
        torch.manual_seed(500 + rank)
        _loss(ddp("train", cur, src)).backward()
        torch.cuda.synchronize()
        avg = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        torch.save({"local": local, "ddp": avg}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ddp_averages_gradients_across_two_ranks(tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(2))
    names = sorted(r0["ddp"])
    assert names == sorted(r1["ddp"]) and len(names) > 500
    enc = [n for n in names if n.startswith(("encoder.", "matching_model."))]
    assert len(enc) > 400, "the encoders' parameters take part in the data-parallel step"
    differ = 0
    scale = max(float(v.abs().max()) for v in r0["ddp"].values())
    for n in names:
        assert torch.equal(r0["ddp"][n], r1["ddp"][n]), f"ranks hold different gradients for {n}"
        l0 = r0["local"].get(n, torch.zeros_like(r0["ddp"][n]))
        l1 = r1["local"].get(n, torch.zeros_like(r0["ddp"][n]))
        want = (l0 + l1) / 2
        # (the sweep backward scatters with fp32 atomics: two runs of the same step differ in the last bits)
        err = float((r0["ddp"][n].double() - want.double()).norm() / max(float(want.double().norm()), 1e-12))
        assert err < 1e-4 or float((r0["ddp"][n] - want).abs().max()) < 1e-6 * scale, (n, err)
        differ += int(not torch.equal(l0, l1))
    assert differ > 0.9 * len(names), "the two ranks saw different data"
