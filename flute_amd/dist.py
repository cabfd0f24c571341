"""Multi-GPU plumbing for the weak-scaling shard: one process per GPU, independent silent-OT instances per rank
(no data-path collective); torch.distributed (RCCL on GPUs, gloo on CPU) only carries the barrier and the
max-over-ranks of the timing. Used by bench.py; covered on CPU by tests/test_dist_gloo.py."""
import os


def env_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, rank, world, device_id=None):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 and not dist.is_initialized():
        kw = {"device_id": device_id} if device_id is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def instance_seed(rank, base=1000):
    """Every rank expands different OT instances: distinct base OTs, deltas and PRG seeds."""
    return base + rank


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(seconds, world, device="cpu"):
    if world <= 1:
        return seconds
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_ots_per_s(n, steps, seconds, world):
    """Whole-job throughput: every rank produced 2n OTs per step (sender + receiver instance)."""
    return world * 2.0 * n * steps / seconds
