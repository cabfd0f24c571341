"""Multi-GPU plumbing for the weak-scaling shard: one process per GPU, independent silent-OT instances per rank.
The expansion itself needs no collective; the one exchange step is the all-gather that reassembles the party's MT
store (the three bit vectors a, b, c of every rank's instance, 6 MiB per rank at n = 2^24) on every GPU --
torch.distributed all_gather_into_tensor = RCCL over xGMI on GPUs, gloo on CPU. The barrier and the max-over-ranks of
the timing go through the same process group. Used by bench.py; covered on CPU by tests/test_dist_gloo.py."""
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


def all_gather_words(local, world):
    """local: 1-D int64 tensor (this rank's a | b | c words). Returns a [world, len] tensor holding every rank's
    words in rank order (on world == 1: the input, reshaped). One collective, no host round trip."""
    import torch
    if world <= 1:
        return local.reshape(1, -1)
    import torch.distributed as dist
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)  # flat: the concatenated form
    dist.all_gather_into_tensor(out, local.contiguous())
    return out.reshape(world, -1)


def gather_mt_store(engine, world, buf=None):
    """The MT bits of this rank's last mt_bits() -> torch tensor on the engine's GPU -> all-gather. Returns
    (gathered [world, 3 * nw] int64 tensor, the reusable local buffer)."""
    import torch
    nw3 = engine.mt_words()
    if buf is None:
        buf = torch.empty(nw3, dtype=torch.int64, device=torch.device("cuda", engine.device))
    engine.mt_bits_to_device(buf.data_ptr(), nw3 * 8)
    return all_gather_words(buf, world), buf
