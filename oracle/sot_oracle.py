"""ctypes binding of the CPU oracle (oracle/libsot_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package (flute_amd) never imports it.

Blocks are numpy uint64 arrays with a trailing dimension of 2 (little endian
u128: [..., 0] = bits 0..63), matching zappot/src/util/block.rs:17-21.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


class Conf(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("n", "scaler", "P", "parts", "size_per", "N2", "N", "depth",
                 "n_blocks", "n64", "cols", "base_ot_count")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class Blk(C.Structure):
    _fields_ = [("w", C.c_uint64 * 2)]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "libsot_oracle.so")
    src = os.path.join(_HERE, "sot_oracle.c")
    need = force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)
    ref_so = os.path.join(_HERE, "_ref", "libbitpolymul_ref.so")
    if os.path.isdir("/root/reference/crates/bitpolymul-sys/bitpolymul2") and not os.path.exists(ref_so):
        need = True
    if need:
        subprocess.check_call(["make", "-C", _HERE, "all"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsot_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        P = C.c_void_p
        u64 = C.c_uint64
        L.orc_configure.argtypes = [u64, u64, u64, C.POINTER(Conf)]
        L.orc_configure.restype = C.c_int
        L.orc_next_prime.argtypes = [u64]
        L.orc_next_prime.restype = u64
        L.orc_aes128_encrypt_blocks.argtypes = [P, P, P, C.c_size_t]
        L.orc_aes_rng_blocks.argtypes = [P, u64, P, C.c_size_t]
        L.orc_cr_hash.argtypes = [P, C.c_size_t]
        L.orc_get_active_path.argtypes = [P, u64]
        L.orc_get_active_path.restype = u64
        L.orc_interleave_point.argtypes = [u64, u64, u64]
        L.orc_interleave_point.restype = u64
        L.orc_pprf_sender_expand.argtypes = [C.POINTER(Conf), P, Blk, Blk, u64, P, P, P]
        L.orc_pprf_sender_expand.restype = C.c_int
        L.orc_pprf_receiver_expand.argtypes = [C.POINTER(Conf), P, P, P, P, P]
        L.orc_pprf_receiver_expand.restype = C.c_int
        L.orc_pprf_points.argtypes = [C.POINTER(Conf), P, P]
        L.orc_transpose_naive.argtypes = [P, P, u64, u64]
        L.orc_transpose_128.argtypes = [P]
        L.orc_gf2x_mul.argtypes = [P, P, P, u64]
        L.orc_gf2x_mul_naive.argtypes = [P, P, P, u64]
        L.orc_set_external_mul.argtypes = [P]
        L.orc_bit_shift_xor.argtypes = [P, u64, P, u64, C.c_uint8]
        L.orc_modp.argtypes = [P, u64, P, u64, u64]
        L.orc_qc_a_poly.argtypes = [C.POINTER(Conf), P]
        L.orc_qc_dual_encode.argtypes = [C.POINTER(Conf), P, P]
        L.orc_qc_dual_encode.restype = C.c_int
        L.orc_qc_dual_encode2.argtypes = [C.POINTER(Conf), P, P, u64, C.c_int, P, P]
        L.orc_qc_dual_encode2.restype = C.c_int
        L.orc_last_cmodp.argtypes = [C.POINTER(u64), C.POINTER(u64)]
        L.orc_last_cmodp.restype = P
        L.orc_sender_hash.argtypes = [Blk, P, u64, P]
        L.orc_receiver_hash.argtypes = [P, u64, P]
        L.orc_mt_bits.argtypes = [P, P, P, u64, P, P, P]
        L.orc_force_portable.argtypes = [C.c_int]
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_get_max_threads.restype = C.c_int
        L.orc_have_aesni.restype = C.c_int
        _LIB = L
    return _LIB


def ref_lib():
    """The reference's own bitpolymul2, compiled from its sources (oracle/_ref). None if absent."""
    global _REF
    if _REF is None:
        so = os.path.join(_HERE, "_ref", "libbitpolymul_ref.so")
        if not os.path.exists(so):
            return None
        _REF = C.CDLL(so)
        _REF.bitpolymul_2_128.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        _REF.bitpolymul_2_64.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    return _REF


def use_reference_multiplier(on=True):
    """Route the oracle's GF(2)[x] product through the reference's bitpolymul_2_128."""
    r = ref_lib() if on else None
    if on and r is None:
        return False
    lib().orc_set_external_mul(C.cast(r.bitpolymul_2_128, C.c_void_p) if on else None)
    return True


def set_num_threads(n):
    """OpenMP team size for the CALLING Python thread's oracle calls (GGM workers, QC rows, transposes)."""
    lib().orc_set_num_threads(int(n))


def max_threads():
    return int(lib().orc_get_max_threads())


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _blk(b):
    b = np.asarray(b, dtype=np.uint64).reshape(2)
    r = Blk()
    r.w[0] = int(b[0])
    r.w[1] = int(b[1])
    return r


def blocks(n):
    return np.zeros((n, 2), dtype=np.uint64)


def u128_to_block(x):
    return np.array([x & 0xFFFFFFFFFFFFFFFF, x >> 64], dtype=np.uint64)


# ---------------------------------------------------------------- config
def configure(n, scaler=2, sec_param=128):
    c = Conf()
    rc = lib().orc_configure(n, scaler, sec_param, C.byref(c))
    if rc != 0:
        raise ValueError("orc_configure failed: %d" % rc)
    return c


# ---------------------------------------------------------------- AES
def aes128_encrypt(key16, data):
    data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    out = np.empty_like(data)
    key = np.frombuffer(bytes(key16), dtype=np.uint8).copy()
    lib().orc_aes128_encrypt_blocks(_p(key), _p(data), _p(out), data.size // 16)
    return out.tobytes()


def aes_rng_blocks(seed_block, first, nblocks):
    seed = np.asarray(seed_block, dtype=np.uint64).reshape(2).copy()
    out = blocks(nblocks)
    lib().orc_aes_rng_blocks(_p(seed), first, _p(out), nblocks)
    return out


def cr_hash(x):
    x = np.ascontiguousarray(x, dtype=np.uint64).copy()
    lib().orc_cr_hash(_p(x), x.shape[0])
    return x


# ---------------------------------------------------------------- PPRF
def get_active_path(bits):
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    return int(lib().orc_get_active_path(_p(bits), bits.size))


def interleave_point(point, tree_idx, total_trees):
    return int(lib().orc_interleave_point(point, tree_idx, total_trees))


def sample_choice_bits(conf, rng):
    """Mirror of pprf.rs:579-620 (InterleavedTransposed) with numpy's rng as bit source."""
    depth, parts = int(conf.depth), int(conf.parts)
    ch = np.zeros((parts, depth), dtype=np.uint8)
    for i in range(parts):
        assert interleave_point(0, i, parts) < conf.N2
        while True:
            ch[i] = rng.integers(0, 2, size=depth, dtype=np.uint8)
            idx = interleave_point(get_active_path(ch[i]), i, parts)
            if idx < conf.N2:
                break
    return ch


def fake_base(conf, rng):
    """Mirror of pprf.rs:858-873: (sender pairs, receiver msgs, choices)."""
    cnt = int(conf.base_ot_count)
    msg2 = rng.integers(0, 2 ** 64, size=(cnt, 2, 2), dtype=np.uint64)
    choices = sample_choice_bits(conf, rng)
    flat = choices.reshape(-1)
    msg = msg2[np.arange(cnt), flat.astype(np.int64)].copy()
    return msg2, msg, choices


def tg_shapes(conf):
    grp = int(conf.parts) // 8
    return (grp, 2, int(conf.depth) - 1, 8, 2), (grp, 8, 4, 2)


def pprf_sender_expand(conf, base_ots, delta, seed, num_threads):
    base_ots = np.ascontiguousarray(base_ots, dtype=np.uint64)
    assert base_ots.shape == (conf.base_ot_count, 2, 2)
    s_shape, l_shape = tg_shapes(conf)
    rT = np.zeros((128, int(conf.cols), 2), dtype=np.uint64)
    sums = np.zeros(s_shape, dtype=np.uint64)
    last = np.zeros(l_shape, dtype=np.uint64)
    rc = lib().orc_pprf_sender_expand(C.byref(conf), _p(base_ots), _blk(delta), _blk(seed), num_threads,
                                      _p(rT), _p(sums), _p(last))
    assert rc == 0
    return rT, sums, last


def pprf_receiver_expand(conf, base_ots, choices, sums, last):
    base_ots = np.ascontiguousarray(base_ots, dtype=np.uint64)
    choices = np.ascontiguousarray(choices, dtype=np.uint8)
    sums = np.ascontiguousarray(sums, dtype=np.uint64)
    last = np.ascontiguousarray(last, dtype=np.uint64)
    rT = np.zeros((128, int(conf.cols), 2), dtype=np.uint64)
    rc = lib().orc_pprf_receiver_expand(C.byref(conf), _p(base_ots), _p(choices), _p(sums), _p(last), _p(rT))
    assert rc == 0
    return rT


def pprf_points(conf, choices):
    choices = np.ascontiguousarray(choices, dtype=np.uint8)
    pts = np.zeros(int(conf.parts), dtype=np.uint64)
    lib().orc_pprf_points(C.byref(conf), _p(choices), _p(pts))
    return pts


# ---------------------------------------------------------------- transpose
def transpose_naive(data, nrows, ncols):
    data = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.uint8))
    out = np.zeros_like(data)
    lib().orc_transpose_naive(_p(data), _p(out), nrows, ncols)
    return out.tobytes()


def transpose_128(m):
    m = np.ascontiguousarray(m, dtype=np.uint64).copy()
    assert m.shape == (128, 2)
    lib().orc_transpose_128(_p(m))
    return m


# ---------------------------------------------------------------- GF(2)[x]
def gf2x_mul(a, b, naive=False):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    assert a.shape == b.shape
    c = np.zeros(2 * a.size, dtype=np.uint64)
    (lib().orc_gf2x_mul_naive if naive else lib().orc_gf2x_mul)(_p(c), _p(a), _p(b), a.size)
    return c


def ref_bitpolymul_2_128(a, b):
    """The reference's own bitpolymul_2_128 (bitpolymul.c:168-279) via oracle/_ref."""
    r = ref_lib()
    n = a.size
    buf = np.zeros(4 * n + 16, dtype=np.uint64)
    off = (-buf.ctypes.data // 8) % 4  # 32-byte alignment
    aa = buf[off:off + n]
    bb = buf[off + n:off + 2 * n]
    cc = buf[off + 2 * n:off + 4 * n]
    aa[:] = a
    bb[:] = b
    r.bitpolymul_2_128(_p(cc), _p(aa), _p(bb), n)
    return cc.copy()


# ---------------------------------------------------------------- QC
def bit_shift_xor(dest, inp, shift):
    dest = np.ascontiguousarray(dest, dtype=np.uint64).copy()
    inp = np.ascontiguousarray(inp, dtype=np.uint64)
    lib().orc_bit_shift_xor(_p(dest), dest.shape[0], _p(inp), inp.shape[0], shift)
    return dest


def modp(dest_len, inp, p):
    inp = np.ascontiguousarray(inp, dtype=np.uint64)
    dest = blocks(dest_len)
    lib().orc_modp(_p(dest), dest_len, _p(inp), inp.shape[0], p)
    return dest


def qc_a_poly(conf):
    a = np.zeros(int(conf.n64), dtype=np.uint64)
    lib().orc_qc_a_poly(C.byref(conf), _p(a))
    return a


def qc_dual_encode(conf, rT):
    rT = np.ascontiguousarray(rT, dtype=np.uint64)
    out = blocks(int(conf.n))
    rc = lib().orc_qc_dual_encode(C.byref(conf), _p(rT), _p(out))
    assert rc == 0
    return out


def qc_dual_encode2(conf, rT, S, packed=True):
    rT = np.ascontiguousarray(rT, dtype=np.uint64)
    S = np.ascontiguousarray(S, dtype=np.uint64)
    out = blocks(int(conf.n))
    Cc = np.zeros(int(conf.n), dtype=np.uint8)
    rc = lib().orc_qc_dual_encode2(C.byref(conf), _p(rT), _p(S), S.size, 1 if packed else 0, _p(out), _p(Cc))
    assert rc == 0
    return out, (None if packed else Cc)


def last_cmodp():
    r = C.c_uint64()
    c = C.c_uint64()
    p = lib().orc_last_cmodp(C.byref(r), C.byref(c))
    n = r.value * c.value * 2
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), shape=(n,))
    return arr.reshape(r.value, c.value, 2).copy()


# ---------------------------------------------------------------- hash + MT
def sender_hash(delta, B):
    B = np.ascontiguousarray(B, dtype=np.uint64)
    out = blocks(2 * B.shape[0])
    lib().orc_sender_hash(_blk(delta), _p(B), B.shape[0], _p(out))
    return out.reshape(B.shape[0], 2, 2)


def receiver_hash(A):
    A = np.ascontiguousarray(A, dtype=np.uint64).copy()
    n = A.shape[0]
    ch = np.zeros((n + 63) // 64, dtype=np.uint64)
    lib().orc_receiver_hash(_p(A), n, _p(ch))
    return A, ch


def mt_bits(send_msgs, recv_msgs, recv_choice_words):
    send_msgs = np.ascontiguousarray(send_msgs, dtype=np.uint64)
    recv_msgs = np.ascontiguousarray(recv_msgs, dtype=np.uint64)
    ch = np.ascontiguousarray(recv_choice_words, dtype=np.uint64)
    n = recv_msgs.shape[0]
    nw = (n + 63) // 64
    a = np.zeros(nw, dtype=np.uint64)
    b = np.zeros(nw, dtype=np.uint64)
    c = np.zeros(nw, dtype=np.uint64)
    lib().orc_mt_bits(_p(send_msgs), _p(recv_msgs), _p(ch), n, _p(a), _p(b), _p(c))
    return a, b, c


# ---------------------------------------------------------------- full pipelines
def random_silent_send(conf, base_ots, delta, seed, num_threads):
    """silent_ot/mod.rs:193-247: returns (msgs [n,2,2], B, rT, tg_sums, tg_last)."""
    rT, sums, last = pprf_sender_expand(conf, base_ots, delta, seed, num_threads)
    B = qc_dual_encode(conf, rT)
    return sender_hash(delta, B), B, rT, sums, last


def random_silent_receive(conf, base_ots, choices, sums, last):
    """silent_ot/mod.rs:391-435: returns (msgs [n,2], choice words, A, rT)."""
    rT = pprf_receiver_expand(conf, base_ots, choices, sums, last)
    S = pprf_points(conf, choices)
    A, _ = qc_dual_encode2(conf, rT, S, packed=True)
    msgs, ch = receiver_hash(A)
    return msgs, ch, A, rT
