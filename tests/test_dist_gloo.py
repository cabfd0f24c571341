"""world_size-2 gloo test of the N>1 plumbing bench.py uses (flute_amd/dist.py): rank-distinct instances,
barrier, max-over-ranks timing, aggregate throughput. CPU only."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from flute_amd import dist as fd
    from flute_amd.silent_ot import QuasiCyclicConf, fake_base
    d = fd.init("gloo", rank, world)
    conf = QuasiCyclicConf.configure(1000)
    rng = np.random.default_rng(fd.instance_seed(rank))
    msg2, _, choices = fake_base(conf, rng)
    fd.barrier(world)
    dt = fd.max_over_ranks(0.5 + rank, world)  # rank 1 is the slow one
    val = fd.aggregate_ots_per_s(1 << 24, 10, dt, world)
    # the one exchange step: all-gather of every rank's a | b | c words (RCCL on GPUs, gloo here)
    import torch
    n = 1000
    nw = (n + 63) // 64
    local = np.random.default_rng(77 + rank).integers(0, 2 ** 63, 3 * nw, dtype=np.int64)
    g = fd.all_gather_words(torch.from_numpy(local), world).numpy()
    q.put((rank, int(msg2[0, 0, 0]), int(choices.sum()), dt, val, g.tobytes()))
    d.destroy_process_group()


def test_two_rank_plumbing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, m0, c0, dt0, v0, g0), (r1, m1, c1, dt1, v1, g1) = res
    # both ranks hold the same gathered store, in rank order, and it unpacks to the concatenated instances
    from flute_amd.mul_triple import MulTriples
    n, nw = 1000, 16
    exp = np.stack([np.random.default_rng(77 + r).integers(0, 2 ** 63, 3 * nw, dtype=np.int64) for r in range(world)])
    assert g0 == g1 == exp.tobytes()
    mts = MulTriples.from_gathered(exp, n)
    assert len(mts) == world * n
    bits = MulTriples._bits(mts.b, len(mts))
    for r in range(world):
        want = MulTriples._bits(exp[r, nw:2 * nw].view(np.uint64), n)
        assert np.array_equal(bits[r * n:(r + 1) * n], want)
    assert (r0, r1) == (0, 1)
    assert m0 != m1  # different instances per rank
    assert dt0 == dt1 == 1.5  # max over ranks
    assert v0 == v1 == 2 * 2.0 * (1 << 24) * 10 / 1.5


def _worker_strong(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    from flute_amd import dist as fd
    d = fd.init("gloo", rank, world)
    # exchange: message q of rank r = the bytes (r, q, i): after the all-to-all message r of rank q holds (r, q, i)
    m = 4096
    send = torch.empty(world * m, dtype=torch.uint8)
    for peer in range(world):
        send[peer * m:(peer + 1) * m] = torch.from_numpy(
            ((np.arange(m) * 7 + 31 * rank + 101 * peer) % 251).astype(np.uint8))
    recv = fd.all_to_all_bytes(send, world)
    ok = all(np.array_equal(recv[src * m:(src + 1) * m].numpy(),
                            ((np.arange(m) * 7 + 31 * src + 101 * rank) % 251).astype(np.uint8)) for src in range(world))
    # MT words of a sharded expansion: [3, chunk / 64] per rank -> [3, world * chunk / 64], vector-major
    nwl = 5
    local = torch.from_numpy((np.arange(3 * nwl, dtype=np.int64) + 1000 * rank).reshape(3, nwl))
    g = fd.gather_sharded_words(local, world).numpy()
    q.put((rank, ok, g.tobytes()))
    d.destroy_process_group()


def test_two_rank_strong_scaling_collectives():
    """all_to_all_bytes (exchange #1 / #2 of one sharded expansion) and gather_sharded_words over gloo, world size 2"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_strong, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    nwl = 5
    want = np.concatenate([(np.arange(3 * nwl, dtype=np.int64) + 1000 * r).reshape(3, nwl) for r in range(world)], axis=1)
    for rank, ok, g in res:
        assert ok, "all-to-all delivered the wrong messages on rank %d" % rank
        assert g == want.tobytes()


def _worker_roles(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    from flute_amd import dist as fd
    d = fd.init("gloo", rank, world)
    nw = 7
    words = np.random.default_rng(5).integers(-2 ** 63, 2 ** 63, (4, nw), dtype=np.int64)  # b, v, a, u
    local = torch.from_numpy(np.concatenate([words[2 * rank], words[2 * rank + 1]]))

    class _Eng:  # RoleSplitPipeline only needs n_out and device from the engine for its buffers
        n_out, device = 64 * nw, 0
    pipe = fd.RoleSplitPipeline(_Eng(), rank, torch_device=torch.device("cpu"))
    pipe.buf.copy_(local)
    got = pipe.join().numpy()
    q.put((rank, got.tobytes()))
    d.destroy_process_group()


def test_two_rank_role_split_join():
    """RoleSplitPipeline.join over gloo, world size 2: all-gather of the two directions' bit vectors and c = a & b ^ u ^ v"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_roles, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    b, v, a, u = np.random.default_rng(5).integers(-2 ** 63, 2 ** 63, (4, 7), dtype=np.int64)
    want = np.stack([a, b, (a & b) ^ u ^ v]).tobytes()
    assert res[0][1] == want and res[1][1] == want
