#!/usr/bin/env python3
"""One-rank RCCL check of exactly the collectives bench.py's N > 1 path issues (dtype, shape and call form), on a box with
a single GPU: all_to_all_single(uint8), all_gather_into_tensor(int64), all_reduce(MAX, float64), barrier."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from flute_amd import dist as fd  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
send = torch.arange(4096, dtype=torch.uint8, device="cuda")
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send)
assert torch.equal(recv, send)
loc = torch.arange(3 * 5, dtype=torch.int64, device="cuda").reshape(3, 5)
flat = torch.empty(15, dtype=torch.int64, device="cuda")
dist.all_gather_into_tensor(flat, loc.contiguous().reshape(-1))
assert torch.equal(flat.reshape(3, 5), loc)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.5
dist.barrier()
torch.cuda.synchronize()
print("rccl one-rank selfcheck ok:", dist.get_backend())
dist.destroy_process_group()
