"""Does RCCL accept two ranks on ONE device?  (python -m torch.distributed.run --nproc-per-node 2 scratch/rccl_one_device.py)"""
import os
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
x = torch.ones(1024, device="cuda:0") * (dist.get_rank() + 1)
try:
    dist.all_reduce(x)
    torch.cuda.synchronize()
    print("rank", dist.get_rank(), "all_reduce ok", x[0].item())
except Exception as e:
    print("rank", dist.get_rank(), "all_reduce failed:", type(e).__name__, str(e)[:200])
