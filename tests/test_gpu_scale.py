"""Full-size GPU checks, bit-exact against the oracle (products through the reference's bitpolymul2, oracle/_ref): n = 2^20,
2^22, ragged sizes, the headline sizes 2^23 / 2^24 (both roles, packed AND unpacked choice bits, hashes, MT words) and
2^25 / 2^26; n = 2^24 also through the size-independent properties of the reference's own tests (check_correlated /
check_random, silent_ot/mod.rs:744-795; MT relation, gmw/src/mul_triple/silent_ot.rs:166-168)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
LSB_CLR = np.uint64(0xFFFFFFFFFFFFFFFE)


def test_2pow20_pipeline_matches_oracle(orc):
    import flute_amd as fa
    n, T = 1 << 20, 4
    conf = orc.configure(n)
    rng = np.random.default_rng(2020)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    orc.use_reference_multiplier(True)  # the reference's bitpolymul2 when oracle/_ref exists (else Karatsuba)
    try:
        o_s, o_B, o_rT, o_sums, o_last = orc.random_silent_send(conf, msg2, delta, seed, T)
        o_r, o_ch, o_A, _ = orc.random_silent_receive(conf, msg, choices, o_sums, o_last)
    finally:
        orc.use_reference_multiplier(False)
    oa, ob, oc = orc.mt_bits(o_s.reshape(-1, 2), o_r, o_ch)
    eng = fa.Engine(n)
    try:
        sums, last = eng.sender_expand(msg2, delta, seed, T)
        assert np.array_equal(sums, o_sums) and np.array_equal(last, o_last)
        assert np.array_equal(eng.download_rT(0), o_rT)
        eng.qc_encode(0)
        assert np.array_equal(eng.download_encoded(0), o_B)
        assert np.array_equal(eng.sender_hash(delta), o_s)
        eng.receiver_expand(msg, choices, sums, last)
        eng.qc_encode(1, eng.points(choices))
        assert np.array_equal(eng.download_encoded(1), o_A)
        r, ch = eng.receiver_hash()
        assert np.array_equal(r, o_r) and np.array_equal(ch, o_ch)
        a, b, c = eng.mt_bits(delta)
        assert np.array_equal(a, oa) and np.array_equal(b, ob) and np.array_equal(c, oc)
    finally:
        eng.close()


@pytest.mark.parametrize("log2n", [24])
def test_full_size_properties(log2n):
    import flute_amd as fa
    n, T = 1 << log2n, 4
    e0, e1 = fa.Engine(n), fa.Engine(n)
    try:
        rng = np.random.default_rng(24)
        x = fa.fake_base(e0.conf, rng)
        y = fa.fake_base(e0.conf, rng)
        d0, d1, s0, s1 = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
        # direction X: party 0 sends (e0 role 0), party 1 receives (e1 role 1); direction Y the other way round
        tg0 = e0.sender_expand(x[0], d0, s0, T)
        e0.qc_encode(0)
        e1.receiver_expand(x[1], x[2], *tg0)
        e1.qc_encode(1, e1.points(x[2]))
        B = e0.download_encoded(0)
        A = e1.download_encoded(1)
        # check_correlated with packed choice bits
        c = (A[:, 0] & np.uint64(1)).astype(bool)
        assert 0.49 < c.mean() < 0.51
        z = A ^ B
        z[:, 0] &= LSB_CLR
        dm = d0.copy()
        dm[0] &= LSB_CLR
        ok = np.where(c, (z[:, 0] == dm[0]) & (z[:, 1] == dm[1]), (z[:, 0] == 0) & (z[:, 1] == 0))
        assert ok.all()
        del z, ok
        # check_random on a window of the hashed outputs
        sm = e0.sender_hash(d0)
        rm, ch = e1.receiver_hash()
        sel = np.unpackbits(ch.view(np.uint8), bitorder="little")[:n].astype(np.int64)
        assert np.array_equal(sel.astype(bool), c)
        idx = np.arange(n)
        assert np.array_equal(sm[idx, sel], rm)
        assert not (sm[idx, 1 - sel] == rm).all(axis=1).any()
        del sm, rm
        # MT relation over all n triples
        tg1 = e1.sender_expand(y[0], d1, s1, T)
        e1.qc_encode(0)
        e0.receiver_expand(y[1], y[2], *tg1)
        e0.qc_encode(1, e0.points(y[2]))
        a0, b0, c0 = e0.mt_bits(d0)
        a1, b1, c1 = e1.mt_bits(d1)
        assert np.array_equal(c0 ^ c1, (a0 ^ a1) & (b0 ^ b1))
        assert 0.49 < np.unpackbits((a0 ^ a1).view(np.uint8)).mean() < 0.51
    finally:
        e0.close()
        e1.close()


@pytest.mark.parametrize("log2n", [21, 22, 23])
def test_fused_conversion_equals_generic_path(log2n, monkeypatch):
    """The fused multi-unit conversion kernels against the one-phase-per-launch generic kernels, bit for bit, at sizes
    where several multi-unit phases exist (check_correlated cannot see an error that sender and receiver share). The
    kernel-selection switches are read once, at sot_ctx_create: the generic path is a second context."""
    import flute_amd as fa
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    out = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("SOT_NO_TOP", "1")
        eng = fa.Engine(n)
        try:
            if not generic:
                msg2, msg, choices = fa.fake_base(eng.conf, rng)
                delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
                seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
            eng.sender_expand(msg2, delta, seed, 4)
            eng.qc_encode(0)
            out.append(eng.download_encoded(0))
        finally:
            eng.close()
    assert np.array_equal(out[0], out[1])


def test_2pow22_encode_matches_oracle_with_reference_multiplier(orc):
    """n = 2^22: QC encode of a random rT against the oracle whose products come from the reference's own
    bitpolymul_2_128 (oracle/_ref)."""
    import flute_amd as fa
    if not orc.use_reference_multiplier(True):
        pytest.skip("oracle/_ref absent")
    try:
        n = 1 << 22
        conf = orc.configure(n)
        rng = np.random.default_rng(22)
        rT = rng.integers(0, 2 ** 64, size=(128, int(conf.cols), 2), dtype=np.uint64)
        want = orc.qc_dual_encode(conf, rT)
    finally:
        orc.use_reference_multiplier(False)
    eng = fa.Engine(n)
    try:
        assert np.array_equal(eng.qc_dual_encode_host(0, rT), want)
    finally:
        eng.close()


def _encode_both_vs_oracle(orc, n, T, seed, receiver=True, hashes=False, unpacked=False):
    """Sender B, receiver A (and with `hashes` the receiver's choice words and the MT words a, b, c of the context's
    two instances; with `unpacked` also ChoiceBitPacking::False: all 128 rows of A and the choice vector C from
    k_sparse_choice, quasi_cyclic_encode.rs:104-117) against the oracle whose products come from the reference's
    bitpolymul2 when oracle/_ref exists."""
    import flute_amd as fa
    conf = orc.configure(n)
    rng = np.random.default_rng(seed)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    prg = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    eng = fa.Engine(n)
    try:
        sums, last = eng.sender_expand(msg2, delta, prg, T)
        eng.qc_encode(0)
        B = eng.download_encoded(0)
        pts = eng.points(choices)
        if receiver:
            eng.receiver_expand(msg, choices, sums, last)
            eng.qc_encode(1, pts)
            A = eng.download_encoded(1)
            if hashes:
                _, ch = eng.receiver_hash(download=False)
                a, b, c = eng.mt_bits(delta)
            if unpacked:
                C_u = eng.qc_encode2(pts, packed=False)
                A_u = eng.download_encoded(1)
    finally:
        eng.close()
    orc.use_reference_multiplier(True)
    try:
        o_rT, o_sums, o_last = orc.pprf_sender_expand(conf, msg2, delta, prg, T)
        assert np.array_equal(sums, o_sums) and np.array_equal(last, o_last)
        o_B = orc.qc_dual_encode(conf, o_rT)
        del o_rT
        assert np.array_equal(B, o_B)
        if receiver:
            o_rT_r = orc.pprf_receiver_expand(conf, msg, choices, o_sums, o_last)
            o_A, _ = orc.qc_dual_encode2(conf, o_rT_r, pts, packed=True)
            if unpacked:
                o_Au, o_Cu = orc.qc_dual_encode2(conf, o_rT_r, pts, packed=False)
                assert np.array_equal(A_u, o_Au), "unpacked A"
                assert np.array_equal(C_u, np.asarray(o_Cu, dtype=np.uint8).reshape(-1)[:n]), "unpacked choice vector C"
                del o_Au, o_Cu
            del o_rT_r
            assert np.array_equal(A, o_A)
            if hashes:
                o_r, o_ch = orc.receiver_hash(o_A)
                assert np.array_equal(ch, o_ch)
                oa, ob, oc = orc.mt_bits(orc.sender_hash(delta, o_B).reshape(-1, 2), o_r, o_ch)
                assert np.array_equal(a, oa) and np.array_equal(b, ob) and np.array_equal(c, oc)
    finally:
        orc.use_reference_multiplier(False)


@pytest.mark.parametrize("n,T", [(2000000, 4), (3 * 2 ** 20 + 12345, 7), (100003, 3)])
def test_odd_sizes_match_oracle(orc, n, T):
    """bench's default --ots 2 000 000 (bench.rs:55-56; P < FFT capacity, no tail terms) and non-power-of-two n."""
    _encode_both_vs_oracle(orc, n, T, n % 9973)


def test_random_sizes_match_oracle(orc):
    """sixteen sizes drawn (seeded) from 2 ... 6 000 000, ten log-uniform and six uniform, with random thread counts: every
    transform size m = 21 ... 24, ragged last tiles, tile counts that are no multiple of the grids -- both roles, choice
    words and MT words against the oracle"""
    rng = np.random.default_rng(20261111)
    sizes = [int(2 ** rng.uniform(1, 22.5)) for _ in range(10)] + [int(rng.integers(1 << 16, 6000000)) for _ in range(6)]
    import flute_amd as fa
    refused = 0
    for i, n in enumerate(sizes):
        T = int(rng.integers(1, 9))
        conf = orc.configure(n)
        if int(conf.n_blocks) * 128 < int(conf.P):
            # about one size in forty: N = N2 / scaler is no multiple of 128 and its last partial Block is needed to hold P
            # bits -- the reference's modp panics on it (quasi_cyclic_encode.rs:289-303), the library refuses the context
            with pytest.raises(fa.SotError, match="unsupported geometry"):
                fa.Engine(n)
            refused += 1
            continue
        _encode_both_vs_oracle(orc, n, T, 7000 + i, receiver=True, hashes=True, unpacked=(i % 3 == 0))
    assert refused <= 2


@pytest.mark.parametrize("log2n", [23, 24])
def test_headline_size_matches_oracle(orc, log2n):
    """n = 2^24 is the size bench.py times (transform size m = 25: k_fft_low<1,16> with nb = 9, k_top<.,9>) and n = 2^23
    is m = 24: both roles, the receiver's choice words and the MT words, bit for bit against the oracle (silent_ot/
    mod.rs:410-482, pprf.rs:456-535). At 2^24 also the unpacked receiver mode (A with all 128 rows, C from k_sparse_choice
    over the full length) against orc.qc_dual_encode2(packed=False)."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref absent")
    _encode_both_vs_oracle(orc, 1 << log2n, 4, 1000 + log2n, receiver=True, hashes=True, unpacked=log2n == 24)


@pytest.mark.parametrize("log2n", [25, 26])
def test_large_sizes_match_oracle(orc, log2n):
    """2^25 and BASELINE.json's largest size 2^26 (m = 26, 27: extra 32-plane passes, whole-unit conversion phases in
    k_conv_wholeunits instead of the top pass), sender AND receiver (k_recv_fix / k_recv_last at depth 20, 21 and the sb
    substitution), choice words and MT words. Needs oracle/_ref for a tolerable oracle run time."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref absent")
    _encode_both_vs_oracle(orc, 1 << log2n, 4, log2n, receiver=True, hashes=True)


def test_unpacked_mode_at_the_headline_size():
    """ChoiceBitPacking::False at n = 2^24 (k_sparse_choice over 120 shifts of a' at full length) through what does not
    need the oracle: check_correlated with the explicit choice vector (silent_ot/mod.rs:744-777), and agreement with the
    packed mode, which the tests above hold bit-exact against the oracle at this size: bit 0 of the packed A is C, the
    other 127 bit rows are the same."""
    import flute_amd as fa
    n, T = 1 << 24, 4
    eng = fa.Engine(n)
    try:
        rng = np.random.default_rng(2424)
        msg2, msg, choices = fa.fake_base(eng.conf, rng)
        delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        snd = fa.Sender.new_with_silent_base_ots(eng, msg2, T)
        rcv = fa.Receiver.new_with_silent_base_ots(eng, msg, choices, T)
        B = snd.correlated_silent_send(delta, seed)
        A, Cv = rcv.correlated_silent_receive(snd.tree_groups, choice_bit_packing=False)
        assert Cv.dtype == np.uint8 and Cv.shape == (n,) and 0.45 * n < int(Cv.sum()) < 0.55 * n
        sel = Cv.astype(bool)
        assert np.array_equal(A[~sel], B[~sel]) and np.array_equal(A[sel], B[sel] ^ delta)
        A_packed = rcv.correlated_silent_receive(snd.tree_groups)
        assert np.array_equal(A_packed[:, 0] & np.uint64(1), Cv.astype(np.uint64))
        assert np.array_equal(A_packed[:, 0] >> np.uint64(1), A[:, 0] >> np.uint64(1))
        assert np.array_equal(A_packed[:, 1], A[:, 1])
    finally:
        eng.close()
