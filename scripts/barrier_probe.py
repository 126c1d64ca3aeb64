"""How long do the end-of-run barrier variants take with one rank on RCCL?  (bench.py's timed region ends with barrier + sync.)"""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
x = torch.randn(8192, 8192, device=dev)
tok = torch.zeros(1, device=dev)
def work():
    y = x
    for _ in range(30):
        y = y @ x * 1e-4
    return y
for name, fn in [("barrier+sync", lambda: (dist.barrier(), torch.cuda.synchronize())),
                 ("sync+barrier+sync", lambda: (torch.cuda.synchronize(), dist.barrier(), torch.cuda.synchronize())),
                 ("all_reduce(token)+sync", lambda: (dist.all_reduce(tok), torch.cuda.synchronize())),
                 ("sync only", lambda: torch.cuda.synchronize())]:
    for rep in range(3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); work(); t1 = time.perf_counter(); fn(); t2 = time.perf_counter()
        if rep:
            print(f"{name:26s} issue {1e3*(t1-t0):7.2f} ms  total {1e3*(t2-t0):8.2f} ms", flush=True)
dist.destroy_process_group()
