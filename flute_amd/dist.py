"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; gloo on CPU).

Strong scaling (default of bench.py --gpus N): ONE expansion sharded over the ranks -- tree sets in the GGM stage, bit
rows of both roles in the QC stage, output ranges in the hash stage (csrc/shard.hip). Two all-to-alls move the leaf
slices and the encoded slices between the stages (ShardedPipeline), an all-gather reassembles the MT words
(gather_sharded_mt). Weak scaling (--mode weak): independent instances per rank, all-gather of the MT store only.
The barrier and the max-over-ranks of the timing go through the same process group. Covered on CPU by
tests/test_dist_gloo.py, on one GPU by tests/test_gpu_shard.py (all ranks in one process, copies instead of RCCL)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# strong scaling: one expansion over all ranks
def all_to_all_bytes(send, world, out=None):
    """send: 1-D uint8 tensor of world equal messages, message q for rank q. Returns the tensor whose message r came from
    rank r (torch.distributed.all_to_all_single). RCCL takes device tensors directly; gloo (rehearsals of GPU ranks on a
    one-GPU box, CPU tests) moves host memory, so device tensors are staged through the host for it."""
    import torch
    import torch.distributed as dist
    if out is None:
        out = torch.empty_like(send)
    if world <= 1:
        out.copy_(send)
        return out
    if send.is_cuda and dist.get_backend() == "gloo":
        h_in, h_out = send.cpu(), torch.empty(send.numel(), dtype=send.dtype)
        dist.all_to_all_single(h_out, h_in)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, send)
    return out


def gather_sharded_words(local, world):
    """local: [3, chunk / 64] int64 words (a | b | c of this rank's outputs). Returns [3, world * chunk / 64] on every
    rank: vector v of the whole expansion = the ranks' words of v in rank order (chunk is a multiple of 64)."""
    import torch
    if world <= 1:
        return local
    import torch.distributed as dist
    flat = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    src = local.contiguous().reshape(-1)  # 1-D in, 1-D out: the concatenated form on every backend
    if local.is_cuda and dist.get_backend() == "gloo":
        h = torch.empty(flat.numel(), dtype=local.dtype)
        dist.all_gather_into_tensor(h, src.cpu())
        flat.copy_(h)
    else:
        dist.all_gather_into_tensor(flat, src)
    return flat.reshape(world, 3, -1).permute(1, 0, 2).reshape(3, -1)


def gather_sharded_blocks(engine, what, world, delta=None, buf=None):
    """The whole output vector of a sharded expansion on every rank (north_star: "RCCL all-gather over xGMI reassembles
    the output vector"; the reference returns it whole: silent_ot/mod.rs:193-211, :391-404). `what`: engine.OUT_B /
    OUT_A (n Blocks), OUT_SENDER_MSGS (n x [Block; 2]) or OUT_RECEIVER_MSGS (n Blocks). Each rank writes its n_local
    outputs into a chunk-sized slot (sot_blocks_to_device; the last rank's slot is ragged), ONE all_gather_into_tensor
    concatenates the slots in rank order, the first n entries are the vector. Returns (int64 tensor [n, 2] or [n, 2, 2]
    viewing the gathered buffer, the reusable local buffer)."""
    import torch
    import torch.distributed as dist
    per = 4 if what == engine.OUT_SENDER_MSGS else 2  # int64 words per output
    chunk = engine.chunk if world > 1 else engine.n_out
    n = engine.conf.n
    if buf is None:
        buf = torch.zeros(chunk * per, dtype=torch.int64, device=torch.device("cuda", engine.device))
    engine.blocks_to_device(what, buf.data_ptr(), buf.numel() * 8, delta=delta)
    if world <= 1:
        flat = buf
    else:
        flat = torch.empty(world * buf.numel(), dtype=buf.dtype, device=buf.device)
        if dist.get_backend() == "gloo":  # rehearsals of GPU ranks over gloo: staged through the host
            h = torch.empty(flat.numel(), dtype=buf.dtype)
            dist.all_gather_into_tensor(h, buf.cpu())
            flat.copy_(h)
        else:
            dist.all_gather_into_tensor(flat, buf)
    out = flat[:n * per]
    return (out.reshape(n, 2, 2) if per == 4 else out.reshape(n, 2)), buf


class ShardedPipeline:
    """One rank's side of a sharded precompute_mts step (gmw/src/mul_triple/silent_ot.rs:79-114 for ONE expansion over
    `world` GPUs). `exchange(send, recv)` is the all-to-all: all_to_all_bytes over torch.distributed by default; the
    single-process tests pass a function that copies between the ranks' buffers instead.

    ordered = True (what bench.py uses): the engine queues its kernels on this pipeline's torch stream
    (sot_ctx_set_stream) and its stage calls no longer wait for the GPU (sot_ctx_set_async); the collectives are issued
    under the same stream, so stages and exchanges are ordered by the stream alone -- one host synchronisation per step
    (the TreeGrp download of the sender's expansion) instead of eight. overlap = True additionally sends the SENDER's
    leaf slices (half of exchange #1, on a second stream) while the receiver's trees expand."""

    def __init__(self, engine, exchange=None, torch_device=None, ordered=False, overlap=False):
        import torch
        self.eng, self.world = engine, engine.world
        dev = torch_device if torch_device is not None else torch.device("cuda", engine.device)
        self.send1 = self.recv1 = None
        if not engine.ggm_replicated:
            self.send1 = torch.empty(engine.world * engine.x1_bytes, dtype=torch.uint8, device=dev)
            self.recv1 = torch.empty_like(self.send1)
        self.send2 = torch.empty(engine.world * engine.x2_bytes, dtype=torch.uint8, device=dev)
        self.recv2 = torch.empty_like(self.send2)
        nwl = engine.chunk // 64
        self.mt = torch.zeros(3, nwl, dtype=torch.int64, device=dev)
        self.exchange = exchange or (lambda s, r: all_to_all_bytes(s, self.world, r))
        self.ordered = bool(ordered) and dev.type == "cuda"
        self.overlap = bool(overlap) and self.ordered and not engine.ggm_replicated
        self.stream = self.stream2 = None
        if self.ordered:
            self.stream, self.stream2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            if self.overlap:  # contiguous staging of one role's halves of the world messages
                h = engine.world * (engine.x1_bytes // 2)
                self.half_s = [torch.empty(h, dtype=torch.uint8, device=dev) for _ in range(2)]
                self.half_r = [torch.empty(h, dtype=torch.uint8, device=dev) for _ in range(2)]
            torch.cuda.synchronize(dev)  # the buffers above were made on the default stream
            engine.set_stream(self.stream.cuda_stream)
            engine.set_async(True)

    def close(self):
        """hand the engine back its own stream (before the torch streams go away)"""
        if self.ordered and self.eng.h:
            self.eng.set_async(False)
            self.eng.set_stream(None)
        self.ordered = False

    def _on_stream(self):
        import contextlib
        import torch
        return torch.cuda.stream(self.stream) if self.ordered else contextlib.nullcontext()

    def expand(self, sender_base, delta, seed, threads, recv_base, recv_choices, tg_sums, tg_last):
        """both directions' GGM stage on this rank's tree sets; returns this rank's TreeGrps (flat arrays, own groups)"""
        with self._on_stream():
            tg = self.eng.sender_expand(sender_base, delta, seed, threads)
            if self.overlap:
                self._send_role(0)  # the sender's slices travel while the receiver's trees expand
            self.eng.receiver_expand(recv_base, recv_choices, tg_sums, tg_last)
        return tg

    def _send_role(self, r):
        """pack role r's halves of the messages of exchange #1 and run their all-to-all: role 0 on the second stream,
        role 1 on the pipeline's own"""
        import torch
        e, w, h = self.eng, self.world, self.eng.x1_bytes // 2
        e.pack_leaves(self.send1.data_ptr(), roles=1 << r)
        st = self.stream2 if r == 0 else self.stream
        if r == 0:
            self.stream2.wait_stream(self.stream)
        with torch.cuda.stream(st):
            self.half_s[r].view(w, h).copy_(self.send1.view(w, 2, h)[:, r])
            self.exchange(self.half_s[r], self.half_r[r])
            self.recv1.view(w, 2, h)[:, r].copy_(self.half_r[r].view(w, h))

    def exchange_leaves(self):
        with self._on_stream():
            if self.eng.ggm_replicated:
                self.eng.local_leaves()
                return
            if self.overlap:
                self._send_role(1)
                self.stream.wait_stream(self.stream2)
            else:
                self.eng.pack_leaves(self.send1.data_ptr())
                self.exchange(self.send1, self.recv1)
                self.sync_exchange()
            self.eng.unpack_leaves(self.recv1.data_ptr())

    def encode(self, points):
        with self._on_stream():
            self.eng.shard_qc_encode(points, self.send2.data_ptr())
            self.exchange(self.send2, self.recv2)
            self.sync_exchange()
            self.eng.unpack_encoded(self.recv2.data_ptr())

    def sync_exchange(self):
        """the collective ran on torch's stream, the engine's kernels run on the context's: order them (nothing to do when
        the context is stream-ordered on the collective's stream)"""
        import torch
        if self.recv2.is_cuda and not self.ordered:
            torch.cuda.current_stream(self.recv2.device).synchronize()

    def mt_bits(self, delta, gather=True):
        """fused hashes + MT bits on this rank's outputs; with `gather`, the whole expansion's a | b | c words
        ([3, world * chunk / 64] int64, the first ceil(n / 64) of each row are the MulTriples vectors) on every rank"""
        with self._on_stream():
            self.eng.mt_bits(delta, download=False)
            nwl = (self.eng.n_out + 63) // 64
            if nwl:
                import torch
                if getattr(self, "_tmp", None) is None or self._tmp.numel() != 3 * nwl:
                    self._tmp = torch.empty(3 * nwl, dtype=torch.int64, device=self.mt.device)  # once: no per-step allocation
                    if self.ordered:
                        torch.cuda.current_stream().synchronize()
                tmp = self._tmp
                self.eng.mt_bits_to_device(tmp.data_ptr(), 3 * nwl * 8)
                self.mt.zero_()
                self.mt[:, :nwl] = tmp.reshape(3, nwl)
            else:
                self.mt.zero_()
            out = gather_sharded_words(self.mt, self.world) if gather else self.mt
        self._hand_over(out)
        return out

    def _hand_over(self, *tensors):
        """results made on the pipeline's stream are about to be used on the caller's: a stream-side wait, no host sync;
        `tensors` came from the caching allocator under the pipeline's stream and must not be recycled there while the
        caller's stream still reads them"""
        if self.ordered:
            import torch
            cur = torch.cuda.current_stream(self.mt.device)
            cur.wait_stream(self.stream)
            for t in tensors:
                if t is not None and t.is_cuda:
                    t.record_stream(cur)

    def gather_blocks(self, what, delta=None, buf=None):
        """gather_sharded_blocks on this pipeline's stream: the whole B / A / hashed-message vector on every rank"""
        with self._on_stream():
            out = gather_sharded_blocks(self.eng, what, self.world, delta=delta, buf=buf)
        self._hand_over(*out)
        return out

    def sync(self):
        """wait for everything this pipeline has queued"""
        if self.ordered:
            self.stream.synchronize()
            self.stream2.synchronize()
        self.eng.sync()


# ---------------------------------------------------------------------------------------------------------------------
# two GPUs: one direction per GPU
def join_role_words(flat, nw):
    """flat: the all-gathered per-direction words of the two ranks, [b | v | a | u] (rank 0 = the sender direction's b, v;
    rank 1 = the receiver direction's a, u; nw words each) -> [3, nw] a | b | c with c = a & b ^ u ^ v
    (gmw/src/mul_triple/silent_ot.rs:93-111: the join of the two directions' bits)."""
    import torch
    b, v, a, u = flat[0:nw], flat[nw:2 * nw], flat[2 * nw:3 * nw], flat[3 * nw:4 * nw]
    return torch.stack([a, b, (a & b) ^ u ^ v])


class RoleSplitPipeline:
    """A party's precompute_mts over TWO GPUs by direction: rank 0 runs the sender instance, rank 1 the receiver instance
    -- the reference runs them concurrently on two thread pools (gmw/src/mul_triple/silent_ot.rs:47,87-91) -- each on an
    ordinary one-GPU context (all trees, all 128 bit rows of its direction). Nothing of the expansion crosses the link:
    each rank reduces its direction to two bit vectors (sot_mt_role_bits_to_device), one all-gather of 2 * n / 8 bytes per
    rank puts both on both GPUs, and the join c = a & b ^ u ^ v is three word-wise tensor operations."""

    def __init__(self, engine, rank, gather=None, torch_device=None):
        import torch
        self.eng, self.rank = engine, rank
        dev = torch_device if torch_device is not None else torch.device("cuda", engine.device)
        self.nw = (engine.n_out + 63) // 64
        self.buf = torch.zeros(2 * self.nw, dtype=torch.int64, device=dev)
        self.gather = gather or self._all_gather

    def _all_gather(self, local):
        import torch
        import torch.distributed as dist
        if getattr(self, "_flat", None) is None:
            self._flat = torch.empty(2 * local.numel(), dtype=local.dtype, device=local.device)  # once
        flat = self._flat
        if local.is_cuda and dist.get_backend() == "gloo":  # rehearsals of GPU ranks over gloo: staged through the host
            h = torch.empty(flat.numel(), dtype=local.dtype)
            dist.all_gather_into_tensor(h, local.cpu())
            flat.copy_(h)
        else:
            dist.all_gather_into_tensor(flat, local)
        return flat

    def direction(self, sender_base, delta, seed, threads, recv_base, recv_choices, tg_sums, tg_last, points):
        """this rank's direction up to its two bit vectors; rank 0 returns the sender's TreeGrps (for the peer party)"""
        e = self.eng
        tg = None
        if self.rank == 0:
            tg = e.sender_expand(sender_base, delta, seed, threads)
            e.qc_encode(0)
            e.mt_role_bits_to_device(0, delta, self.buf.data_ptr(), self.buf.numel() * 8)
        else:
            e.receiver_expand(recv_base, recv_choices, tg_sums, tg_last)
            e.qc_encode(1, points)
            e.mt_role_bits_to_device(1, None, self.buf.data_ptr(), self.buf.numel() * 8)
        return tg

    def join(self):
        """-> [3, ceil(n / 64)] int64 a | b | c, the party's whole MT store, on both GPUs"""
        return join_role_words(self.gather(self.buf), self.nw)

