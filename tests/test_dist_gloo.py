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
    q.put((rank, int(msg2[0, 0, 0]), int(choices.sum()), dt, val))
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
    (r0, m0, c0, dt0, v0), (r1, m1, c1, dt1, v1) = res
    assert (r0, r1) == (0, 1)
    assert m0 != m1  # different instances per rank
    assert dt0 == dt1 == 1.5  # max over ranks
    assert v0 == v1 == 2 * 2.0 * (1 << 24) * 10 / 1.5
