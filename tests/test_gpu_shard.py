"""ONE expansion sharded over 2, 4 and 8 ranks (csrc/shard.hip, include/flute_sot.h sot_shard_*), all ranks as contexts
of one process on one GPU, the two all-to-alls as device copies between the ranks' buffers: the gathered TreeGrps, B,
A, hashed messages, choice words and MT words must be BYTE-IDENTICAL to the one-GPU context's for the same seeds (and so
to the oracle's, which test_gpu_parity / test_gpu_scale pin the one-GPU context to)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flute_amd
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests need an MI355X")
    return flute_amd


def _local_exchange(pipes):
    """all-to-all among pipelines that live in this process: message q of rank r's send buffer -> message r of rank q's
    receive buffer. Called once per rank like the collective; the copies happen when the last rank has called."""
    state = {"calls": 0}

    def make(r):
        def exchange(send, recv):
            state.setdefault("send", {})[r] = send
            state.setdefault("recv", {})[r] = recv
        return exchange

    def flush():
        import torch
        w = len(pipes)
        send, recv = state.pop("send"), state.pop("recv")
        m = send[0].numel() // w
        for r in range(w):
            for q in range(w):
                recv[q][r * m:(r + 1) * m].copy_(send[r][q * m:(q + 1) * m])
        torch.cuda.synchronize()
    return make, flush


def _run_sharded(fa, n, T, world, replicated, x, y, tg_y, delta, seed, scaler=2):
    from flute_amd import dist as fd
    engs = [fa.ShardedEngine(n, r, world, ggm_replicated=replicated, mult_type=fa.MultType(scaler)) for r in range(world)]
    try:
        pipes = [None] * world
        make, flush = _local_exchange(pipes)
        for r, e in enumerate(engs):
            pipes[r] = fd.ShardedPipeline(e, exchange=make(r))
        conf = engs[0].conf
        s_shape, l_shape = conf.treegrp_shapes()
        tg_sums, tg_last = np.zeros(s_shape, np.uint64), np.zeros(l_shape, np.uint64)
        for r, p in enumerate(pipes):
            sums, last = p.expand(x[0], delta, seed, T, y[1], y[2], *tg_y)
            g = list(engs[r].groups) if not replicated else list(range(conf.parts // 8))
            tg_sums[g], tg_last[g] = sums[g], last[g]
            if not replicated:  # a sharded rank leaves the other groups' entries alone
                others = [i for i in range(conf.parts // 8) if i not in g]
                assert not sums[others].any() and not last[others].any()
        pts = engs[0].points(y[2])
        # exchange #1: every rank packs, then the copies, then every rank unpacks
        if replicated:
            for p in pipes:
                p.exchange_leaves()
        else:
            for p in pipes:
                p.eng.pack_leaves(p.send1.data_ptr())
                p.exchange(p.send1, p.recv1)
            flush()
            for p in pipes:
                p.eng.unpack_leaves(p.recv1.data_ptr())
        for p in pipes:
            p.eng.shard_qc_encode(pts, p.send2.data_ptr())
            p.exchange(p.send2, p.recv2)
        flush()
        out = {"B": [], "A": [], "s": [], "r": [], "ch": [], "mt": []}
        for p in pipes:
            p.eng.unpack_encoded(p.recv2.data_ptr())
            out["B"].append(p.eng.download_encoded(0))
            out["A"].append(p.eng.download_encoded(1))
            out["s"].append(p.eng.sender_hash(delta))
            rm, ch = p.eng.receiver_hash()
            out["r"].append(rm)
            out["ch"].append(np.concatenate([ch, np.zeros(p.eng.chunk // 64 - ch.size, np.uint64)]))
            out["mt"].append(p.mt_bits(delta, gather=False).cpu().numpy().view(np.uint64))
        assert sum(e.n_out for e in engs) == n and all(e.k0 == r * e.chunk for r, e in enumerate(engs))
        nw = (n + 63) // 64
        mt = np.concatenate(out["mt"], axis=1)[:, :nw]
        return (tg_sums, tg_last, np.concatenate(out["B"]), np.concatenate(out["A"]), np.concatenate(out["s"]),
                np.concatenate(out["r"]), np.concatenate(out["ch"])[:nw], mt)
    finally:
        for e in engs:
            e.close()


CASES = [(1000, 2, 2, False), (1000, 2, 4, False), (1000, 2, 8, False), (1000, 1, 2, True), (1000, 3, 8, True),
         (40000, 4, 4, False), (65536, 5, 8, False), (129, 3, 2, False), (1 << 20, 4, 2, False), (1 << 20, 4, 8, False),
         (1 << 20, 4, 4, True), (3 * 2 ** 20 + 12345, 7, 4, False),
         # the row-half / slice skipping of 4 and 8 ranks on every low-kernel template (m = 22 ... 24) and a ragged size
         (100003, 3, 8, True), (1 << 21, 4, 4, False), (1 << 22, 4, 8, False), (1 << 23, 4, 4, False),
         (3 * 2 ** 20 + 12345, 5, 8, False),
         (1 << 24, 4, 4, False),  # m = 25: the persistent low kernel (k_fft_low_p) on one row half (64 live rows)
         # every BASELINE config-3 shape: 2^24 over 2 ranks (replicated GGM: what bench.py --ggm auto picks there) and over
         # 8 ranks, 2^26 over 4 and 8 ranks (m = 27: 32-plane passes, k_conv_wholeunits on 64 / 32 live rows)
         (1 << 24, 4, 2, True), (1 << 24, 4, 8, False), (1 << 26, 4, 4, False), (1 << 26, 4, 8, False)]


@pytest.mark.parametrize("n,T,world,replicated,scaler", [(40000, 3, 4, False, 3), (100003, 2, 8, False, 4), (65536, 2, 2, True, 3)])
def test_sharded_expansion_other_scalers(fa, n, T, world, replicated, scaler):
    """MultType::QuasiCyclic{scaler: 3 / 4} sharded: one product pass per public polynomial a_s over the slot Blocks"""
    test_sharded_expansion_equals_one_gpu(fa, n, T, world, replicated, scaler)


@pytest.mark.parametrize("n,T,world,replicated", CASES)
def test_sharded_expansion_equals_one_gpu(fa, n, T, world, replicated, scaler=2):
    rng = np.random.default_rng(n % 1009 + world)
    ref = fa.Engine(n, mult_type=fa.MultType(scaler))
    try:
        conf = ref.conf
        x = fa.fake_base(conf, rng)  # direction X: this party sends
        y = fa.fake_base(conf, rng)  # direction Y: this party receives
        delta, seed, dp, sp = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
        tg_y = ref.sender_expand(y[0], dp, sp, T)  # the peer's TreeGrps for direction Y
        w_sums, w_last = ref.sender_expand(x[0], delta, seed, T)
        ref.qc_encode(0)
        ref.receiver_expand(y[1], y[2], *tg_y)
        ref.qc_encode(1, ref.points(y[2]))
        w_B, w_A = ref.download_encoded(0), ref.download_encoded(1)
        w_s = ref.sender_hash(delta)
        w_r, w_ch = ref.receiver_hash()
        w_mt = np.stack(ref.mt_bits(delta))
    finally:
        ref.close()
    sums, last, B, A, sm, rm, ch, mt = _run_sharded(fa, n, T, world, replicated, x, y, tg_y, delta, seed, scaler)
    assert np.array_equal(sums, w_sums) and np.array_equal(last, w_last), "TreeGrps"
    assert np.array_equal(B, w_B), "sender B"
    assert np.array_equal(A, w_A), "receiver A"
    assert np.array_equal(sm, w_s) and np.array_equal(rm, w_r) and np.array_equal(ch, w_ch), "hashes"
    assert np.array_equal(mt, w_mt), "MT words"


def test_sharded_expansion_random_shapes(fa, orc):
    """twelve seeded random (n, threads, ranks, GGM mode, scaler) shapes, n from 2 to 3 000 000: byte-identical to one GPU"""
    rng = np.random.default_rng(424242)
    done = 0
    while done < 12:
        n = int(2 ** rng.uniform(1, 21.5))
        T, world = int(rng.integers(1, 9)), int(rng.choice([2, 4, 8]))
        replicated, scaler = bool(rng.integers(0, 2)), int(rng.choice([2, 2, 2, 3]))
        conf = orc.configure(n, scaler)
        if int(conf.n_blocks) * 128 < int(conf.P):
            continue  # a geometry the reference's modp panics on: refused by sot_ctx_create (test_gpu_scale.py)
        test_sharded_expansion_equals_one_gpu(fa, n, T, world, replicated, scaler)
        done += 1


def test_sharded_context_refuses_whole_matrix_stages(fa):
    from flute_amd.lib import SotError
    e = fa.ShardedEngine(1000, 1, 2)
    try:
        rng = np.random.default_rng(1)
        x = fa.fake_base(e.conf, rng)
        d, s = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
        e.sender_expand(x[0], d, s, 2)
        with pytest.raises(SotError):
            e.qc_encode(0)  # the leaves of the other rank's tree sets are not here
        with pytest.raises(SotError):
            e.download_rT(0)
        with pytest.raises(SotError):
            e.local_leaves()  # not a replicated GGM stage
        import torch
        buf = torch.zeros(e.world * e.x2_bytes, dtype=torch.uint8, device="cuda:0")
        with pytest.raises(SotError):
            e.shard_qc_encode(e.points(x[2]), buf.data_ptr())  # no leaf slices yet
    finally:
        e.close()
    with pytest.raises(SotError):
        fa.ShardedEngine(1000, 0, 3)
    with pytest.raises(SotError):
        fa.ShardedEngine(1000, 2, 2)


@pytest.mark.parametrize("n,T", [(1000, 2), (100003, 3), (1 << 20, 4), (1 << 24, 4)])
def test_role_split_equals_one_gpu(fa, n, T):
    """two GPUs by direction (flute_amd/dist.py::RoleSplitPipeline, sot_mt_role_bits_to_device): the sender instance on
    one context, the receiver instance on another, joined from their bit vectors = the MT words of one context"""
    import torch
    from flute_amd import dist as fd
    rng = np.random.default_rng(n)
    ref = fa.Engine(n)
    conf = ref.conf
    x = fa.fake_base(conf, rng)
    y = fa.fake_base(conf, rng)
    delta, seed, delta_peer, seed_peer = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    try:
        tg_y = ref.sender_expand(y[0], delta_peer, seed_peer, T)
        tg_x = ref.sender_expand(x[0], delta, seed, T)
        ref.qc_encode(0)
        ref.receiver_expand(y[1], y[2], *tg_y)
        pts = ref.points(y[2])
        ref.qc_encode(1, pts)
        want = np.stack(ref.mt_bits(delta))
    finally:
        ref.close()
    engs = [fa.Engine(n), fa.Engine(n)]
    try:
        pipes = [fd.RoleSplitPipeline(e, r, gather=lambda local: None) for r, e in enumerate(engs)]
        tg = pipes[0].direction(x[0], delta, seed, T, y[1], y[2], tg_y[0], tg_y[1], pts)
        assert np.array_equal(tg[0], tg_x[0]) and np.array_equal(tg[1], tg_x[1])
        assert pipes[1].direction(x[0], delta, seed, T, y[1], y[2], tg_y[0], tg_y[1], pts) is None
        torch.cuda.synchronize()
        flat = torch.cat([pipes[0].buf, pipes[1].buf])  # what the all-gather leaves on both ranks
        got = fd.join_role_words(flat, pipes[0].nw).cpu().numpy().view(np.uint64)
        assert np.array_equal(got, want)
        from flute_amd.lib import SotError
        with pytest.raises(SotError):  # the receiver direction was never encoded on context 0
            engs[0].mt_role_bits_to_device(1, None, pipes[0].buf.data_ptr(), pipes[0].buf.numel() * 8)
    finally:
        for e in engs:
            e.close()
