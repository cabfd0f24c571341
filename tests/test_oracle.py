"""Pins the CPU oracle against every known-answer vector and property the
reference's own tests hold for the silent-OT hot path,
plus FIPS-197 and the reference's C multiplier (oracle/_ref)."""
import ctypes as C

import numpy as np
import pytest

M64 = (1 << 64) - 1


def blk_to_int(b):
    return int(b[0]) | (int(b[1]) << 64)


def ints(bl):
    return [blk_to_int(b) for b in bl.reshape(-1, 2)]


# ------------------------------------------------------------------ AES
def test_aes_fips197_kat(orc):
    key = bytes(range(16))
    pt = bytes.fromhex("00112233445566778899aabbccddeeff")
    ct = bytes.fromhex("69c4e0d86a7b0430d8cdb78070b4c55a")
    assert orc.aes128_encrypt(key, pt) == ct
    # FIPS-197 appendix B
    key = bytes.fromhex("2b7e151628aed2a6abf7158809cf4f3c")
    pt = bytes.fromhex("3243f6a8885a308d313198a2e0370734")
    ct = bytes.fromhex("3925841d02dc09fbdc118597196a0b32")
    assert orc.aes128_encrypt(key, pt) == ct


def test_aes_portable_matches_aesni(orc):
    rng = np.random.default_rng(1)
    key = rng.integers(0, 256, 16, dtype=np.uint8).tobytes()
    data = rng.integers(0, 256, 16 * 37, dtype=np.uint8).tobytes()
    fast = orc.aes128_encrypt(key, data)
    orc.lib().orc_force_portable(1)
    try:
        slow = orc.aes128_encrypt(key, data)
    finally:
        orc.lib().orc_force_portable(0)
    assert fast == slow


def test_fixed_keys_are_the_reference_constants(orc):
    # pprf.rs:675-688 and aes_hash.rs:59-66: u128 constants -> LE key bytes
    k0 = (91389970179024809574621370423327856399).to_bytes(16, "little")
    k1 = (297966570818470707816499469807199042980).to_bytes(16, "little")
    kh = (193502124791825095790518994062991136444).to_bytes(16, "little")
    assert k0.hex() == "0f173c468f95ba6b9cac20a86911c144"
    assert k1.hex() == "a4bdc17a018bc8916ab3db802c462ae0"
    assert kh.hex() == "bc96179a4d75fcd215491e3555279391"
    x = np.array([[5, 7]], dtype=np.uint64)
    want = bytes(a ^ b for a, b in zip(orc.aes128_encrypt(kh, x.tobytes()), x.tobytes()))
    assert orc.cr_hash(x).tobytes() == want


def test_aes_rng_is_ctr_keystream(orc):
    # aes_rng.rs:109-136: block j = AES_seed(LE128(j))
    seed = np.array([0x1122334455667788, 0x99AABBCCDDEEFF00], dtype=np.uint64)
    ks = orc.aes_rng_blocks(seed, 3, 5)
    for j in range(5):
        ctr = (3 + j).to_bytes(16, "little")
        assert ks[j].tobytes() == orc.aes128_encrypt(seed.tobytes(), ctr)


# ------------------------------------------------------------------ config
@pytest.mark.parametrize("n,exp", [
    (1000, dict(P=16411, parts=128, size_per=264, depth=9, N2=33792, N=16896, n_blocks=132, n64=264, cols=264,
                base_ot_count=1152)),
    (2000000, dict(P=2000003, parts=120, size_per=33336, depth=16, N2=4000320, N=2000160, n_blocks=15626,
                   n64=31252, cols=31253, base_ot_count=1920)),
    (1 << 20, dict(P=1048583, parts=120, size_per=17480, depth=15, N2=2097600, N=1048800, n_blocks=8193,
                   n64=16386, cols=16388, base_ot_count=1800)),
    (1 << 24, dict(P=16777259, parts=120, size_per=279624, depth=19, N2=33554880, N=16777440, n_blocks=131073,
                   n64=262146, cols=262148, base_ot_count=2280)),
    (1 << 26, dict(P=67108879, parts=120, size_per=1118488, depth=21, N2=134218560, N=67109280, n_blocks=524291,
                   n64=1048582, cols=1048583, base_ot_count=2520)),
])
def test_configure_table(orc, n, exp):
    # QuasiCyclicConf::configure quasi_cyclic_encode.rs:177-193 (values: SURVEY.md section 8)
    d = orc.configure(n).as_dict()
    for k, v in exp.items():
        assert d[k] == v, (k, d[k], v)


def test_next_prime_is_strictly_greater(orc):
    assert orc.lib().orc_next_prime(16384) == 16411
    assert orc.lib().orc_next_prime(16411) == 16417
    assert orc.lib().orc_next_prime(2) == 3


# ------------------------------------------------------------------ transpose
def test_transpose_sse_rust_kat(orc):
    # util/transpose.rs:168-210
    data = bytes([0b00000001, 0b00000000] * 16)
    expected = bytes([0xFF, 0xFF] + [0, 0] * 15)
    assert orc.transpose_naive(data, 16, 16) == expected


def test_transpose_128_matches_naive(orc):
    rng = np.random.default_rng(2)
    m = rng.integers(0, 2 ** 64, size=(128, 2), dtype=np.uint64)
    fast = orc.transpose_128(m)
    assert fast.tobytes() == orc.transpose_naive(m.tobytes(), 128, 128)
    assert orc.transpose_128(fast).tobytes() == m.tobytes()


# ------------------------------------------------------------------ QC pieces
def test_basic_bit_shift_xor(orc):
    # silent_ot/mod.rs:797-806
    dest = np.zeros((2, 2), dtype=np.uint64)
    inp = np.full((2, 2), M64, dtype=np.uint64)
    out = orc.bit_shift_xor(dest, inp, 10)
    assert blk_to_int(out[0]) == (1 << 128) - 1
    assert blk_to_int(out[1]) == ((1 << 128) - 1) >> 10


def _bits_of(blocks):
    v = 0
    for i, b in enumerate(blocks.reshape(-1, 2)):
        v |= blk_to_int(b) << (128 * i)
    return v


def _to_blocks(v, nblocks):
    out = np.zeros((nblocks, 2), dtype=np.uint64)
    for i in range(nblocks):
        out[i, 0] = (v >> (128 * i)) & M64
        out[i, 1] = (v >> (128 * i + 64)) & M64
    return out


def test_basic_modp(orc):
    # silent_ot/mod.rs:808-834: 1026 one-bits folded mod p = 223
    i_bits, p = 1026, 223
    n = (p + 127) // 128
    inp = _to_blocks((1 << i_bits) - 1, (i_bits + 127) // 128)
    dv = (1 << p) - 1
    c = (i_bits + p - 1) // p
    for j in range(1, c):
        rem = min(p, i_bits - j * p)
        dv ^= (1 << rem) - 1
    dest = orc.modp(n, inp, p)
    assert _bits_of(dest) & ((1 << p) - 1) == dv


@pytest.mark.parametrize("nblk,p", [(264, 16411), (22, 1283), (10, 521)])
def test_modp_equals_mathematical_fold(orc, nblk, p):
    rng = np.random.default_rng(3)
    inp = rng.integers(0, 2 ** 64, size=(nblk, 2), dtype=np.uint64)
    dest_len = nblk // 2
    assert dest_len * 128 >= p
    got = _bits_of(orc.modp(dest_len, inp, p))
    v = _bits_of(inp)
    want = 0
    while v:
        want ^= v & ((1 << p) - 1)
        v >>= p
    assert got == want


# ------------------------------------------------------------------ GF(2)[x]
@pytest.mark.parametrize("n64", [1, 7, 33, 264, 300, 1000])
def test_gf2x_mul_matches_naive(orc, n64):
    rng = np.random.default_rng(n64)
    a = rng.integers(0, 2 ** 64, n64, dtype=np.uint64)
    b = rng.integers(0, 2 ** 64, n64, dtype=np.uint64)
    want = orc.gf2x_mul(a, b, naive=True)
    assert np.array_equal(orc.gf2x_mul(a, b), want)
    orc.lib().orc_force_portable(1)
    try:
        if n64 <= 300:
            assert np.array_equal(orc.gf2x_mul(a, b), want)
    finally:
        orc.lib().orc_force_portable(0)


@pytest.mark.parametrize("n64", [264, 300, 1000, 16386])
def test_gf2x_mul_matches_reference_bitpolymul(orc, n64):
    """Our product == the reference's own bitpolymul_2_128 (bitpolymul.c:168-279), compiled from
    its sources into oracle/_ref. n64 = 264 / 16386 are the FFT operand sizes at n=1000 / 2^20."""
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference absent)")
    rng = np.random.default_rng(n64 + 1)
    a = rng.integers(0, 2 ** 64, n64, dtype=np.uint64)
    b = rng.integers(0, 2 ** 64, n64, dtype=np.uint64)
    assert np.array_equal(orc.gf2x_mul(a, b), orc.ref_bitpolymul_2_128(a, b))


# ------------------------------------------------------------------ PPRF property (pprf.rs:875-930)
def _custom_pprf_conf(orc, domain, pnt):
    c = orc.Conf()
    c.size_per = domain
    c.parts = pnt
    c.depth = max(1, (domain - 1).bit_length())
    c.N2 = domain * pnt
    c.cols = (pnt * domain + 127) // 128
    c.base_ot_count = c.depth * pnt
    return c


@pytest.mark.parametrize("threads", [1, 3])
def test_silent_pprf_property(orc, threads):
    conf = _custom_pprf_conf(orc, 334, 5 * 8)
    rng = np.random.default_rng(42)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = np.array([M64, M64], dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    s_out, sums, last = orc.pprf_sender_expand(conf, msg2, delta, seed, threads)
    r_out = orc.pprf_receiver_expand(conf, msg, choices, sums, last)
    points = set(int(p) for p in orc.pprf_points(conf, choices))
    assert len(points) == 40
    out = s_out ^ r_out
    out_t = orc.transpose_naive(out.tobytes(), 128, int(conf.cols) * 128)
    out_t = np.frombuffer(out_t, dtype=np.uint64).reshape(-1, 2)
    ones = (out_t[:, 0] == M64) & (out_t[:, 1] == M64)
    zeros = (out_t[:, 0] == 0) & (out_t[:, 1] == 0)
    idx = set(np.nonzero(ones)[0].tolist())
    assert idx == points
    assert int(ones.sum() + zeros.sum()) == out_t.shape[0]


def test_thread_count_changes_sender_output_but_not_correlation(orc):
    conf = _custom_pprf_conf(orc, 334, 16)
    rng = np.random.default_rng(7)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    s1, su1, la1 = orc.pprf_sender_expand(conf, msg2, delta, seed, 1)
    s2, su2, la2 = orc.pprf_sender_expand(conf, msg2, delta, seed, 2)
    assert not np.array_equal(s1, s2)  # roots come from AesRng(seed ^ t): pprf.rs:115
    # group 0 belongs to worker 0 in both cases and takes keystream blocks 0..7
    assert np.array_equal(su1[0], su2[0]) and np.array_equal(la1[0], la2[0])


# ------------------------------------------------------------------ integration (mod.rs:836-904)
def _run_pair(orc, n, threads, seed_int, packed):
    conf = orc.configure(n)
    rng = np.random.default_rng(seed_int)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    rT_s, sums, last = orc.pprf_sender_expand(conf, msg2, delta, seed, threads)
    B = orc.qc_dual_encode(conf, rT_s)
    rT_r = orc.pprf_receiver_expand(conf, msg, choices, sums, last)
    S = orc.pprf_points(conf, choices)
    A, Cc = orc.qc_dual_encode2(conf, rT_r, S, packed=packed)
    return conf, delta, A, B, Cc


def test_correlated_silent_ot_unpacked(orc):
    # check_correlated with explicit choice vector (mod.rs:744-777, 836-872)
    conf, delta, A, B, Cc = _run_pair(orc, 1000, 2, 42, packed=False)
    d = blk_to_int(delta)
    a, b = ints(A), ints(B)
    assert 0 < int(Cc.sum()) < 1000
    for i in range(1000):
        eq = [a[i] == b[i], a[i] == b[i] ^ d]
        assert eq[Cc[i]] and not eq[Cc[i] ^ 1], i


def test_correlated_silent_ot_packed(orc):
    conf, delta, A, B, _ = _run_pair(orc, 1000, 2, 43, packed=True)
    mask = ((1 << 128) - 1) ^ 1
    d = blk_to_int(delta)
    for i, (x, y) in enumerate(zip(ints(A), ints(B))):
        c = x & 1
        eq = [(x & mask) == (y & mask), (x & mask) == ((y ^ d) & mask)]
        assert eq[c] and not eq[c ^ 1], i


def test_random_silent_ot_and_mt_relation(orc):
    """check_random (mod.rs:779-795) and c1^c2 = (a1^a2)(b1^b2) (gmw/src/mul_triple/silent_ot.rs:166-168)."""
    n, T = 1000, 2
    conf = orc.configure(n)
    rng = np.random.default_rng(44)
    parties = []
    # direction X: party 0 sends, party 1 receives; direction Y: party 1 sends, party 0 receives
    runs = []
    for _ in range(2):
        msg2, msg, choices = orc.fake_base(conf, rng)
        delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
        smsgs, B, rT, sums, last = orc.random_silent_send(conf, msg2, delta, seed, T)
        rmsgs, ch, A, _ = orc.random_silent_receive(conf, msg, choices, sums, last)
        for i in range(n):
            c = (int(ch[i // 64]) >> (i % 64)) & 1
            m = [smsgs[i, 0].tobytes(), smsgs[i, 1].tobytes()]
            assert rmsgs[i].tobytes() == m[c] and rmsgs[i].tobytes() != m[c ^ 1]
        runs.append((smsgs, rmsgs, ch))
    # party 0: sender of run 0, receiver of run 1; party 1: sender of run 1, receiver of run 0
    a0, b0, c0 = orc.mt_bits(runs[0][0].reshape(-1, 2), runs[1][1], runs[1][2])
    a1, b1, c1 = orc.mt_bits(runs[1][0].reshape(-1, 2), runs[0][1], runs[0][2])
    nw = (n + 63) // 64
    tail = np.uint64((1 << (n % 64)) - 1) if n % 64 else np.uint64(M64)
    left = c0 ^ c1
    right = (a0 ^ a1) & (b0 ^ b1)
    left[nw - 1] &= tail
    right[nw - 1] &= tail
    assert np.array_equal(left, right)


def test_qc_encode_equals_block_domain_formula(orc):
    """B[k] = V[k] ^ V[k+P] ^ sum_{m = k mod P} (a * V1)[m] with V the untransposed leaf vector
    (SURVEY.md appendix B): the formulation the HIP path computes. Checked on bit-rows 0, 5, 127."""
    n = 1000
    conf = orc.configure(n)
    rng = np.random.default_rng(45)
    rT = rng.integers(0, 2 ** 64, size=(128, int(conf.cols), 2), dtype=np.uint64)
    B = ints(orc.qc_dual_encode(conf, rT))
    Vb = orc.transpose_naive(rT.tobytes(), 128, int(conf.cols) * 128)
    V = [blk_to_int(b) for b in np.frombuffer(Vb, dtype=np.uint64).reshape(-1, 2)]
    a = orc.qc_a_poly(conf)
    nb128 = int(conf.n_blocks) * 128
    P = int(conf.P)
    for r in (0, 5, 127):
        v0 = sum(((V[c] >> r) & 1) << c for c in range(nb128))
        v1 = sum(((V[nb128 + c] >> r) & 1) << c for c in range(nb128))
        b1 = np.frombuffer(v1.to_bytes(nb128 // 8, "little"), dtype=np.uint64)
        t = int.from_bytes(orc.gf2x_mul(a, b1).tobytes(), "little") ^ v0
        folded = 0
        while t:
            folded ^= t & ((1 << P) - 1)
            t >>= P
        for k in range(n):
            assert (B[k] >> r) & 1 == (folded >> k) & 1, (r, k)
