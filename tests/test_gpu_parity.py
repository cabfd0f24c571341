"""GPU parity tests: every stage of the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs. Integer/bit work: the bar is bit-exact equality."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

M64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def fa():
    import flute_amd
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests need an MI355X")
    return flute_amd


def _inputs(orc, n, seed):
    conf = orc.configure(n)
    rng = np.random.default_rng(seed)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    prg = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    return conf, msg2, msg, choices, delta, prg


# ------------------------------------------------------------------ fixtures-level entry points
def test_aes_kat_on_gpu(fa, orc):
    import ctypes as C
    L = fa.load_library()
    key = np.frombuffer(bytes(range(16)), dtype=np.uint8).copy()
    pt = np.frombuffer(bytes.fromhex("00112233445566778899aabbccddeeff"), dtype=np.uint8).copy()
    out = np.zeros(16, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.sot_aes128_encrypt(0, p(key), p(pt), p(out), 1) == 0
    assert out.tobytes().hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"  # FIPS-197 C.1
    rng = np.random.default_rng(0)
    data = rng.integers(0, 2 ** 64, size=(5000, 2), dtype=np.uint64)
    k2 = rng.integers(0, 256, 16, dtype=np.uint8)
    o2 = np.zeros_like(data)
    assert L.sot_aes128_encrypt(0, p(k2), p(data), p(o2), 5000) == 0
    assert o2.tobytes() == orc.aes128_encrypt(k2.tobytes(), data.tobytes())
    o3 = np.zeros_like(data)
    assert L.sot_cr_hash(0, p(data), p(o3), 5000) == 0
    assert np.array_equal(o3, orc.cr_hash(data))


def test_transpose128_on_gpu(fa, orc):
    import ctypes as C
    L = fa.load_library()
    rng = np.random.default_rng(1)
    tiles = rng.integers(0, 2 ** 64, size=(37, 128, 2), dtype=np.uint64)
    out = np.zeros_like(tiles)
    assert L.sot_transpose128(0, tiles.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), 37) == 0
    for i in range(37):
        assert np.array_equal(out[i], orc.transpose_128(tiles[i])), i
    # reference KAT (util/transpose.rs:168-210) embedded in a 128x128 tile: column 0 set -> row 0 all ones
    t = np.zeros((1, 128, 2), dtype=np.uint64)
    t[0, :, 0] = 1
    o = np.zeros_like(t)
    assert L.sot_transpose128(0, t.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p), 1) == 0
    assert o[0, 0, 0] == M64 and o[0, 0, 1] == M64 and not o[0, 1:].any()


# ------------------------------------------------------------------ stage by stage
@pytest.mark.parametrize("n,T,seed", [(1000, 2, 42), (1000, 1, 7), (40000, 4, 3), (1 << 16, 5, 11), (1, 2, 5), (129, 3, 6)])
def test_pipeline_stages_match_oracle(fa, orc, n, T, seed):
    conf, msg2, msg, choices, delta, prg = _inputs(orc, n, seed)
    eng = fa.Engine(n)
    try:
        assert eng.conf.c.as_dict() == conf.as_dict()
        # --- PPRF sender
        o_rT_s, o_sums, o_last = orc.pprf_sender_expand(conf, msg2, delta, prg, T)
        sums, last = eng.sender_expand(msg2, delta, prg, T)
        assert np.array_equal(sums, o_sums), "TreeGrp sums"
        assert np.array_equal(last, o_last), "TreeGrp last_ots"
        assert np.array_equal(eng.download_rT(0), o_rT_s), "sender rT"
        # --- PPRF receiver
        o_rT_r = orc.pprf_receiver_expand(conf, msg, choices, o_sums, o_last)
        eng.receiver_expand(msg, choices, sums, last)
        assert np.array_equal(eng.download_rT(1), o_rT_r), "receiver rT"
        pts = eng.points(choices)
        assert np.array_equal(pts, orc.pprf_points(conf, choices))
        # --- QC encode
        o_B = orc.qc_dual_encode(conf, o_rT_s)
        eng.qc_encode(0)
        B = eng.download_encoded(0)
        assert np.array_equal(B, o_B), "sender dual_encode"
        o_A, _ = orc.qc_dual_encode2(conf, o_rT_r, pts, packed=True)
        eng.qc_encode(1, pts)
        A = eng.download_encoded(1)
        assert np.array_equal(A, o_A), "receiver dual_encode2"
        # --- hashes
        o_msgs = orc.sender_hash(delta, o_B)
        assert np.array_equal(eng.sender_hash(delta), o_msgs)
        o_rmsgs, o_ch = orc.receiver_hash(o_A)
        rmsgs, ch = eng.receiver_hash()
        assert np.array_equal(rmsgs, o_rmsgs) and np.array_equal(ch, o_ch)
        # --- MT bits (this context's sender and receiver instance as one party's two directions)
        oa, ob, oc = orc.mt_bits(o_msgs.reshape(-1, 2), o_rmsgs, o_ch)
        a, b, c = eng.mt_bits(delta)
        assert np.array_equal(a, oa) and np.array_equal(b, ob) and np.array_equal(c, oc)
        # --- reference properties on the GPU outputs (check_random, silent_ot/mod.rs:779-795)
        sel = np.unpackbits(ch.view(np.uint8), bitorder="little")[:n].astype(np.int64)
        msgs = eng.sender_hash(delta)
        chosen = msgs[np.arange(n), sel]
        other = msgs[np.arange(n), 1 - sel]
        assert np.array_equal(chosen, rmsgs)
        assert not (other == rmsgs).all(axis=1).any()
    finally:
        eng.close()


def test_dual_encode_from_host_rT(fa, orc):
    n = 5000
    conf = orc.configure(n)
    rng = np.random.default_rng(9)
    rT = rng.integers(0, 2 ** 64, size=(128, int(conf.cols), 2), dtype=np.uint64)
    eng = fa.Engine(n)
    try:
        assert np.array_equal(eng.qc_dual_encode_host(0, rT), orc.qc_dual_encode(conf, rT))
        S = np.sort(rng.choice(int(conf.N2), size=int(conf.parts), replace=False)).astype(np.uint64)
        want, _ = orc.qc_dual_encode2(conf, rT, S, packed=True)
        assert np.array_equal(eng.qc_dual_encode_host(1, rT, S), want)
    finally:
        eng.close()


def test_mirror_classes_random_ot(fa, orc):
    """random_silent_ot of the reference (silent_ot/mod.rs:874-904) through the mirrored Sender/Receiver."""
    n, T = 1000, 2
    eng = fa.Engine(n)
    try:
        rng = np.random.default_rng(42)
        msg2, msg, choices = fa.fake_base(eng.conf, rng)
        delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        snd = fa.Sender.new_with_silent_base_ots(eng, msg2, T)
        rcv = fa.Receiver.new_with_silent_base_ots(eng, msg, choices, T)
        s_out = snd.random_silent_send(delta, seed)
        r_out, ch = rcv.random_silent_receive(snd.tree_groups)
        sel = np.unpackbits(ch.view(np.uint8), bitorder="little")[:n].astype(np.int64)
        assert 0 < sel.sum() < n
        assert np.array_equal(s_out[np.arange(n), sel], r_out)
        assert not (s_out[np.arange(n), 1 - sel] == r_out).all(axis=1).any()
    finally:
        eng.close()


def test_tree_groups_through_wire_codec(fa, orc):
    """Sender's TreeGrps -> pprf::Msg::TreeGrp bytes (flute_amd/wire.py) -> shuffled arrival (pprf.rs:128: sender
    threads emit them in any order) -> receiver: the OT relation still holds."""
    from flute_amd import wire
    n, T = 40000, 4
    eng = fa.Engine(n)
    try:
        rng = np.random.default_rng(9)
        msg2, msg, choices = fa.fake_base(eng.conf, rng)
        delta, seed = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
        snd = fa.Sender.new_with_silent_base_ots(eng, msg2, T)
        rcv = fa.Receiver.new_with_silent_base_ots(eng, msg, choices, T)
        s_out = snd.random_silent_send(delta, seed)
        msgs = wire.tree_grps_to_msgs(eng.conf, *snd.tree_groups)
        assert len(msgs) == eng.conf.parts // 8
        msgs = [msgs[i] for i in rng.permutation(len(msgs))]
        tg = wire.msgs_to_tree_grps(eng.conf, msgs)
        assert all(np.array_equal(a, b) for a, b in zip(tg, snd.tree_groups))
        r_out, ch = rcv.random_silent_receive(tg)
        sel = np.unpackbits(ch.view(np.uint8), bitorder="little")[:n].astype(np.int64)
        assert np.array_equal(s_out[np.arange(n), sel], r_out)
    finally:
        eng.close()


def test_two_party_mt_relation(fa, orc):
    """c1 ^ c2 == (a1 ^ a2) & (b1 ^ b2) (gmw/src/mul_triple/silent_ot.rs:166-168) with two providers."""
    n, T = 3000, 3
    e0, e1 = fa.Engine(n), fa.Engine(n)
    try:
        rng = np.random.default_rng(77)
        x = fa.fake_base(e0.conf, rng)  # direction X: party 0 sends, party 1 receives
        y = fa.fake_base(e0.conf, rng)  # direction Y: party 1 sends, party 0 receives
        p0 = fa.SilentMtProvider(e0, x[0], y[1], y[2], T)
        p1 = fa.SilentMtProvider(e1, y[0], x[1], x[2], T)
        d0, d1, s0, s1 = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
        tg0 = p0.send_phase(d0, s0)
        tg1 = p1.send_phase(d1, s1)
        p0.receive_phase(tg1)
        p1.receive_phase(tg0)
        m0, m1 = p0.request_mts(1000), p1.request_mts(1000)
        assert p0.mts_available() == n - 1000
        bits = fa.MulTriples._bits
        a = bits(m0.a, 1000) ^ bits(m1.a, 1000)
        b = bits(m0.b, 1000) ^ bits(m1.b, 1000)
        c = bits(m0.c, 1000) ^ bits(m1.c, 1000)
        assert np.array_equal(c, a & b)
        assert 0 < a.sum() < 1000 and 0 < b.sum() < 1000
    finally:
        e0.close()
        e1.close()


def test_mt_store_gather_single_rank(fa, orc):
    """sot_mt_bits_to_device + dist.gather_mt_store (world 1: the device copy, no collective) hand over exactly the
    a | b | c words that sot_mt_bits downloads; before mt_bits the call is an ESTATE error."""
    import torch
    from flute_amd import dist as fd
    from flute_amd.lib import SotError
    n, T = 3000, 2
    e = fa.Engine(n)
    try:
        rng = np.random.default_rng(5)
        x = fa.fake_base(e.conf, rng)
        y = fa.fake_base(e.conf, rng)
        delta, seed, dp, sp = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
        tg_y = e.sender_expand(y[0], dp, sp, T)
        e.sender_expand(x[0], delta, seed, T)
        e.qc_encode(0)
        e.receiver_expand(y[1], y[2], *tg_y)
        e.qc_encode(1, e.points(y[2]))
        buf = torch.zeros(e.mt_words(), dtype=torch.int64, device="cuda:0")
        with pytest.raises(SotError):
            e.mt_bits_to_device(buf.data_ptr(), buf.numel() * 8)
        a, b, c = e.mt_bits(delta)
        with pytest.raises(SotError):
            e.mt_bits_to_device(buf.data_ptr(), 8)  # too small
        g, _ = fd.gather_mt_store(e, 1, buf)
        torch.cuda.synchronize()
        got = g.cpu().numpy().view(np.uint64)
        assert got.shape == (1, e.mt_words())
        assert np.array_equal(got[0], np.concatenate([a, b, c]))
        mts = fa.MulTriples.from_gathered(got, n)
        assert len(mts) == n and np.array_equal(mts.c, c)
    finally:
        e.close()


def test_encode_leaves_receiver_rT_untouched(fa, orc):
    """The receiver's noise vector sb (silent_ot/mod.rs:560-574) replaces bit row 0 only on the encode path's loads:
    the expanded matrix itself (pprf::Receiver::expand's return value) is still intact after dual_encode2."""
    n, T = 1000, 2
    e = fa.Engine(n)
    try:
        rng = np.random.default_rng(9)
        y = fa.fake_base(e.conf, rng)
        d, s = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
        tg = e.sender_expand(y[0], d, s, T)
        e.receiver_expand(y[1], y[2], *tg)
        before = e.download_rT(1)
        e.qc_encode(1, e.points(y[2]))
        assert np.array_equal(before, e.download_rT(1))
    finally:
        e.close()


def test_calls_out_of_order_are_errors_not_garbage(fa, orc):
    """The C ABI keeps the pipeline state per role: a stage called before the one that feeds it returns an error
    (SotError from the mirror) instead of reading stale or unwritten HBM; after the error the context still works."""
    from flute_amd.lib import SotError
    n, T = 1000, 2
    e = fa.Engine(n)
    try:
        rng = np.random.default_rng(3)
        msg2, msg, choices = fa.fake_base(e.conf, rng)
        delta, seed = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
        with pytest.raises(SotError):
            e.qc_encode(0)  # nothing expanded yet
        with pytest.raises(SotError):
            e.sender_hash(delta)  # nothing encoded yet
        tg = e.sender_expand(msg2, delta, seed, T)
        with pytest.raises(SotError):
            e.sender_hash(delta)  # expanded, not encoded
        with pytest.raises(SotError):
            e.mt_bits(delta)  # needs both roles encoded
        with pytest.raises(ValueError):
            e.receiver_expand(msg, choices, tg[0][:-1], tg[1])  # TreeGrp arrays of the wrong shape
        with pytest.raises(SotError):
            e.sender_expand(msg2, delta, seed, 0)  # num_threads = 0
        # the context is still usable: the full sequence gives the oracle's result
        e.qc_encode(0)
        s_out = e.sender_hash(delta)
        e.receiver_expand(msg, choices, *tg)
        e.qc_encode(1, e.points(choices))
        r_out, ch = e.receiver_hash()
        sel = np.unpackbits(ch.view(np.uint8), bitorder="little")[:n].astype(np.int64)
        assert np.array_equal(s_out[np.arange(n), sel], r_out)
    finally:
        e.close()


def test_two_host_threads_on_one_context(fa, orc):
    """The reference runs both directions of one SilentMtProvider concurrently (tokio::join! of random_silent_send and
    random_silent_receive, gmw/src/mul_triple/silent_ot.rs:87-91). Sender and receiver share GGM scratch, the transform
    buffers and one stream, so every entry point takes the context's lock: two threads calling sot_random_silent_send
    and sot_random_silent_receive on ONE context (ctypes releases the GIL) each get the oracle's bytes, run after run."""
    import threading
    n, T = 40000, 4
    conf = orc.configure(n)
    rng = np.random.default_rng(17)
    x = orc.fake_base(conf, rng)  # direction X: this context sends
    y = orc.fake_base(conf, rng)  # direction Y: this context receives
    delta, seed, dp, sp = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    o_s, _, _, _, _ = orc.random_silent_send(conf, x[0], delta, seed, T)
    _, y_sums, y_last = orc.pprf_sender_expand(conf, y[0], dp, sp, T)
    o_r, o_ch, _, _ = orc.random_silent_receive(conf, y[1], y[2], y_sums, y_last)
    oa, ob, oc = orc.mt_bits(o_s.reshape(-1, 2), o_r, o_ch)
    eng = fa.Engine(n)
    try:
        snd = fa.Sender.new_with_silent_base_ots(eng, x[0], T)
        rcv = fa.Receiver.new_with_silent_base_ots(eng, y[1], y[2], T)
        for _ in range(3):
            res, errs = {}, []

            def send():
                try:
                    res["s"] = snd.random_silent_send(delta, seed)
                except Exception as e:  # surfaced below
                    errs.append(e)

            def recv():
                try:
                    res["r"] = rcv.random_silent_receive((y_sums, y_last))
                except Exception as e:
                    errs.append(e)

            th = [threading.Thread(target=send), threading.Thread(target=recv)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errs, errs
            assert np.array_equal(res["s"], o_s)
            assert np.array_equal(res["r"][0], o_r) and np.array_equal(res["r"][1], o_ch)
            a, b, c = eng.mt_bits(delta)
            assert np.array_equal(a, oa) and np.array_equal(b, ob) and np.array_equal(c, oc)
    finally:
        eng.close()


def test_receiver_hash_leaves_the_mt_store_alone(fa, orc):
    """sot_receiver_hash keeps its choice words in a buffer of its own: the a | b | c store of an earlier sot_mt_bits is
    still what sot_mt_bits_to_device hands over; uploading new leaves (sot_qc_dual_encode_host) invalidates the store."""
    import torch
    from flute_amd.lib import SotError
    n, T = 3000, 2
    e = fa.Engine(n)
    try:
        rng = np.random.default_rng(15)
        x = fa.fake_base(e.conf, rng)
        delta, seed = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
        tg = e.sender_expand(x[0], delta, seed, T)
        e.qc_encode(0)
        e.receiver_expand(x[1], x[2], *tg)
        e.qc_encode(1, e.points(x[2]))
        a, b, c = e.mt_bits(delta)
        _, ch = e.receiver_hash()
        assert np.array_equal(ch, a)  # a = the receiver's choice bits
        buf = torch.zeros(e.mt_words(), dtype=torch.int64, device="cuda:0")
        e.mt_bits_to_device(buf.data_ptr(), buf.numel() * 8)
        assert np.array_equal(buf.cpu().numpy().view(np.uint64), np.concatenate([a, b, c]))
        rT = rng.integers(0, 2 ** 64, size=(128, e.conf.cols, 2), dtype=np.uint64)
        e.qc_dual_encode_host(1, rT, e.points(x[2]))
        with pytest.raises(SotError):
            e.mt_bits_to_device(buf.data_ptr(), buf.numel() * 8)
    finally:
        e.close()


@pytest.mark.parametrize("n,T,seed", [(1000, 2, 42), (40000, 3, 8), (1 << 20, 4, 20)])
def test_receiver_unpacked_choice_bits(fa, orc, n, T, seed):
    """ChoiceBitPacking::False (quasi_cyclic_encode.rs:65-68,104-117), the mode of the reference's own
    correlated_silent_ot test (silent_ot/mod.rs:836-872): A keeps all 128 rows of rT, the choice vector C is a separate
    product. Bit-exact against the oracle, then check_correlated with the explicit choice vector (mod.rs:744-777: no
    masking of bit 0), and the hash of an unpacked A is refused."""
    from flute_amd.lib import SotError
    conf, msg2, msg, choices, delta, prg = _inputs(orc, n, seed)
    delta = np.array([M64, M64], dtype=np.uint64)  # Block::all_ones() as in the reference test
    eng = fa.Engine(n)
    try:
        snd = fa.Sender.new_with_silent_base_ots(eng, msg2, T)
        rcv = fa.Receiver.new_with_silent_base_ots(eng, msg, choices, T)
        B = snd.correlated_silent_send(delta, prg)
        A, Cv = rcv.correlated_silent_receive(snd.tree_groups, choice_bit_packing=False)
        o_rT_s, o_sums, o_last = orc.pprf_sender_expand(conf, msg2, delta, prg, T)
        o_rT_r = orc.pprf_receiver_expand(conf, msg, choices, o_sums, o_last)
        o_A, o_C = orc.qc_dual_encode2(conf, o_rT_r, rcv.S, packed=False)
        assert np.array_equal(B, orc.qc_dual_encode(conf, o_rT_s))
        assert np.array_equal(A, o_A), "unpacked A"
        assert Cv.dtype == np.uint8 and np.array_equal(Cv, o_C), "choice vector C"
        assert 0 < Cv.sum() < n
        # check_correlated, choice = Some(C): A == B ^ C * delta on all 128 bits
        want = B.copy()
        want[Cv.astype(bool)] ^= delta
        assert np.array_equal(A, want)
        with pytest.raises(SotError):
            eng.receiver_hash()
        with pytest.raises(SotError):
            eng.mt_bits(delta)
        # the packed mode on the same leaves: bit 0 of A is C, the other 127 rows are the same
        A_packed = rcv.correlated_silent_receive(snd.tree_groups)
        assert np.array_equal(A_packed[:, 0] & np.uint64(1), Cv.astype(np.uint64))
        assert np.array_equal(A_packed[:, 0] >> np.uint64(1), A[:, 0] >> np.uint64(1)) and np.array_equal(A_packed[:, 1], A[:, 1])
        eng.receiver_hash()  # allowed again
    finally:
        eng.close()


@pytest.mark.parametrize("scaler,n,T", [(3, 40000, 3), (3, 1 << 16, 4), (4, 1000, 2), (4, 100003, 5), (3, 1 << 20, 4)])
def test_other_scalers_match_oracle(fa, orc, scaler, n, T):
    """MultType::QuasiCyclic{scaler} for scaler 3 and 4 (quasi_cyclic_encode.rs:127-138: scaler - 1 public polynomials a_s
    from AesRng(Block::from(s)); :262-286: c = b0 + sum_s a_s b_s): the whole pipeline of both roles -- TreeGrps, rT, B, A,
    hashes, MT words, and the unpacked choice vector (points in segment s contribute shifts of a_s') -- bit for bit against
    the oracle, at sizes whose geometry the reference accepts (n_blocks * 128 >= P; its modp panics otherwise)."""
    conf = orc.configure(n, scaler=scaler)
    assert conf.n_blocks * 128 >= conf.P
    rng = np.random.default_rng(100 * scaler + T)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    prg = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    o_rT, o_sums, o_last = orc.pprf_sender_expand(conf, msg2, delta, prg, T)
    o_B = orc.qc_dual_encode(conf, o_rT)
    o_rTr = orc.pprf_receiver_expand(conf, msg, choices, o_sums, o_last)
    pts = orc.pprf_points(conf, choices)
    o_A, _ = orc.qc_dual_encode2(conf, o_rTr, pts, packed=True)
    o_Au, o_C = orc.qc_dual_encode2(conf, o_rTr, pts, packed=False)
    eng = fa.Engine(n, mult_type=fa.MultType(scaler))
    try:
        assert eng.conf.c.as_dict() == conf.as_dict()
        sums, last = eng.sender_expand(msg2, delta, prg, T)
        assert np.array_equal(sums, o_sums) and np.array_equal(last, o_last)
        assert np.array_equal(eng.download_rT(0), o_rT)
        eng.qc_encode(0)
        assert np.array_equal(eng.download_encoded(0), o_B), "sender B"
        eng.receiver_expand(msg, choices, sums, last)
        assert np.array_equal(eng.points(choices), pts)
        assert np.array_equal(eng.download_rT(1), o_rTr)
        eng.qc_encode(1, pts)
        assert np.array_equal(eng.download_encoded(1), o_A), "receiver A (packed)"
        s_msgs = eng.sender_hash(delta)
        r_msgs, ch = eng.receiver_hash()
        o_s = orc.sender_hash(delta, o_B)
        o_r, o_ch = orc.receiver_hash(o_A)
        assert np.array_equal(s_msgs, o_s.reshape(s_msgs.shape)) and np.array_equal(r_msgs, o_r) and np.array_equal(ch, o_ch)
        a, b, c = eng.mt_bits(delta)
        oa, ob, oc = orc.mt_bits(o_s.reshape(-1, 2), o_r, o_ch)
        assert np.array_equal(a, oa) and np.array_equal(b, ob) and np.array_equal(c, oc)
        Cv = eng.qc_encode2(pts, packed=False)
        assert np.array_equal(eng.download_encoded(1), o_Au), "receiver A (unpacked)"
        assert np.array_equal(Cv, np.asarray(o_C, dtype=np.uint8).reshape(-1)[:n]), "choice vector C"
    finally:
        eng.close()


# ------------------------------------------------------------------ device-resident outputs, stream-ordered context
def test_blocks_to_device_and_stream_ordered_context(fa, orc):
    """sot_blocks_to_device hands out the same B / A / hashed messages as the host downloads; a context queued on the
    caller's stream without per-stage waits (sot_ctx_set_stream / sot_ctx_set_async) produces the same MT words as the
    synchronous call sequence (whose results the oracle pins in test_pipeline_stages_match_oracle)."""
    import torch
    n, T = 40000, 3
    conf, msg2, msg, choices, delta, prg = _inputs(orc, n, 21)
    eng = fa.Engine(n)
    dev = torch.device("cuda", 0)
    try:
        buf = torch.zeros(n * 4, dtype=torch.int64, device=dev)
        with pytest.raises(fa.SotError):  # nothing encoded yet
            eng.blocks_to_device(eng.OUT_B, buf.data_ptr(), buf.numel() * 8)
        sums, last = eng.sender_expand(msg2, delta, prg, T)
        eng.receiver_expand(msg, choices, sums, last)
        pts = eng.points(choices)
        eng.qc_encode(0)
        eng.qc_encode(1, pts)
        B, A = eng.download_encoded(0), eng.download_encoded(1)
        smsgs = eng.sender_hash(delta)
        rmsgs, _ = eng.receiver_hash()
        want_mt = eng.mt_bits(delta)
        for what, want, per in ((eng.OUT_B, B, 2), (eng.OUT_A, A, 2), (eng.OUT_SENDER_MSGS, smsgs, 4),
                                (eng.OUT_RECEIVER_MSGS, rmsgs, 2)):
            buf.fill_(-1)
            torch.cuda.synchronize()
            eng.blocks_to_device(what, buf.data_ptr(), buf.numel() * 8, delta=delta)
            got = buf[:n * per].cpu().numpy().view(np.uint64).reshape(want.shape)
            assert np.array_equal(got, want), what
            assert (buf[n * per:] == -1).all()  # nothing written past the n outputs
        with pytest.raises(fa.SotError):  # destination too small
            eng.blocks_to_device(eng.OUT_SENDER_MSGS, buf.data_ptr(), n * 32 - 8, delta=delta)
        with pytest.raises(fa.SotError):  # the sender's hash needs delta
            eng.blocks_to_device(eng.OUT_SENDER_MSGS, buf.data_ptr(), buf.numel() * 8)
        with pytest.raises(fa.SotError):
            eng.blocks_to_device(7, buf.data_ptr(), buf.numel() * 8)
        # the same step queued on a torch stream, no wait between the stages that hand nothing to the host
        st = torch.cuda.Stream(dev)
        nw = (n + 63) // 64
        mt = torch.zeros(3 * nw, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        eng.set_stream(st.cuda_stream)
        eng.set_async(True)
        with torch.cuda.stream(st):
            for rep in range(2):  # the second pass reuses the pinned staging ring
                sums2, last2 = eng.sender_expand(msg2, delta, prg, T)
                eng.receiver_expand(msg, choices, sums2, last2)
                eng.qc_encode(0)
                eng.qc_encode(1, pts)
                eng.mt_bits(delta, download=False)
                eng.mt_bits_to_device(mt.data_ptr(), mt.numel() * 8)
        st.synchronize()
        assert np.array_equal(sums2, sums) and np.array_equal(last2, last)
        got = mt.cpu().numpy().view(np.uint64).reshape(3, nw)
        assert all(np.array_equal(got[i], want_mt[i]) for i in range(3))
        eng.set_async(False)
        eng.set_stream(None)
        a, b, c = eng.mt_bits(delta)  # back on its own stream, synchronous
        assert np.array_equal(a, want_mt[0]) and np.array_equal(c, want_mt[2])
    finally:
        eng.close()
