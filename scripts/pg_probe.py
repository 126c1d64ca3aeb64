"""Does an initialised RCCL process group by itself change the step time?  (20 steps of hero_cfg3, cuda sync only.)"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import bench_workloads
mode = sys.argv[1]
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
if mode != "none":
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", device_id=dev)
    if mode == "used":
        dist.barrier()
wl = bench_workloads.WORKLOADS["hero_cfg3"](dev, 0)
with torch.inference_mode():
    for _ in range(3): wl.step()
    torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(20): wl.step(i)
        torch.cuda.synchronize()
        print(mode, f"{(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
if mode != "none":
    dist.destroy_process_group()
