#!/usr/bin/env python3
"""One-rank RCCL rehearsal of every collective dist.py / bench.py issue (a 1-GPU box cannot hold two RCCL ranks):
init with device_id, barrier, all_reduce(MAX / SUM), gather to rank 0, reduce.  Exit status 0 = all of them ran."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group(backend="nccl", device_id=dev)
dist.barrier()
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
a = torch.arange(12, dtype=torch.float64, device=dev).reshape(2, 2, 3)
parts = [torch.empty_like(a)]
dist.gather(a, parts, dst=0)
assert torch.equal(parts[0], a)
dist.reduce(a, dst=0, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("nccl api ok", float(t.item()))
sys.exit(0)
