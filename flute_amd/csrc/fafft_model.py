"""Executable model of the Frobenius-pruned additive FFT ("FAFFT") used by qc.hip, in the exact in-place
word layout of the HIP kernels, for ONE bit row. Design aid / spec; never imported by the product.

Idea (Li-Chen-Chang-Cheng-Yang, "Frobenius additive FFT", restated for the Cantor basis over nimbers):
a polynomial f in GF(2)[x] with deg f < 2^m is determined by its values on the coset beta_m + W_m of
GF(2^32); squaring maps the coset to itself (index j -> j ^ (j >> 1)) with orbits of size 32, and
f(w^2) = f(w)^2, so one value per orbit suffices. The set {j : bits m-1, m-2, m-4, m-8, m-16 of j are 0}
is a cross-section, i.e. the LCH butterfly network can drop the "bit = 1" branch at those five levels.
At a pruned level r = m - i = 2^t the surviving branch is x + tau*y with x, y in GF(2^(2^t)) and
tau = u_t + tau_lo (u_t = nimber 2^(2^t), tau_lo in GF(2^(2^t))): the result is (x + tau_lo*y) | y << 2^t, so
pruning doubles the width of the values in place and the data stays at 2^m bits from start to end:

  Z[w], w < 2^(m-5): 32-bit words of the row; bit p of Z[w] is position 32 w + p.
  level r < m - 4 acts on word-index bit b(r) = m - r - 5; levels r = m-4..m on the bit index inside words.
  after the pruned level r = 2^t, word-index bit b(2^t) is bit t of the *plane* (nimber coordinate) index.
"""
import sys
import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__file__))
from gf2fft_model import nim_mul, point, conversion_phases  # noqa: E402

PRUNED = (1, 2, 4, 8, 16)


def rtab():
    """R[t][k] = row k (bitmask over input bits j) of 'multiply by 2^t'."""
    R = [[0] * 32 for _ in range(32)]
    for t in range(32):
        for j in range(32):
            v = nim_mul(1 << t, 1 << j)
            for k in range(32):
                if (v >> k) & 1:
                    R[t][k] |= 1 << j
    return R


_R = None


def mat_rows(c):
    """rows[k] = bitmask over j: output coordinate k of c (x) y takes input coordinate j."""
    global _R
    if _R is None:
        _R = rtab()
    rows = [0] * 32
    t = 0
    while c:
        if c & 1:
            for k in range(32):
                rows[k] ^= _R[t][k]
        c >>= 1
        t += 1
    return rows


class Geom:
    def __init__(self, m):
        assert 21 <= m <= 30
        self.m = m
        self.nw = 1 << (m - 5)
        self.pm = sum(1 << (m - r) for r in PRUNED)  # absolute-position mask of the pruned bits
        self.plane_bit = [m - r - 5 for r in PRUNED]  # word-index bit of plane-index bit t

    def plane_off(self, k):
        o = 0
        for t in range(5):
            if (k >> t) & 1:
                o |= 1 << self.plane_bit[t]
        return o

    def q(self, J, i):
        """twiddle index of the butterfly block holding absolute position J at level i."""
        return ((1 << self.m) | (J & ~self.pm)) >> i


def apply_rows(rows, K, y):
    """t[k] = XOR_j y[j] & -(rows[k]>>j & 1), k, j < K. y: list of numpy arrays / ints."""
    out = []
    for k in range(K):
        acc = 0
        for j in range(K):
            if (rows[k] >> j) & 1:
                acc = acc ^ y[j]
        out.append(acc)
    return out


def level_word(Z, g, r, inverse):
    """word-level r (i = m - r >= 5), in place on the uint32 array Z."""
    m = g.m
    i = m - r
    b = i - 5
    npr = sum(1 for p in PRUNED if p < r)
    K = 1 << npr
    pruned = r in PRUNED
    planes = [g.plane_off(k) for k in range(K)]
    pmask_bits = sum(1 << g.plane_bit[t] for t in range(npr))
    for w in range(g.nw):
        if w & pmask_bits or (w >> b) & 1:
            continue
        tau = point(g.q(w << 5, i))
        if pruned:
            assert tau >> r == 1
            tau &= (1 << r) - 1
        rows = mat_rows(tau)
        xs = [w | p for p in planes]
        ys = [w | p | (1 << b) for p in planes]
        x = [int(Z[a]) for a in xs]
        y = [int(Z[a]) for a in ys]
        if pruned:
            t = apply_rows(rows, K, y)
            for k in range(K):
                Z[xs[k]] = x[k] ^ t[k]
            continue
        if inverse:
            y = [a ^ c for a, c in zip(y, x)]
        t = apply_rows(rows, K, y)
        x = [a ^ c for a, c in zip(x, t)]
        if not inverse:
            y = [a ^ c for a, c in zip(y, x)]
        for k in range(K):
            Z[xs[k]] = x[k]
            Z[ys[k]] = y[k]


LOWMASK = (0x55555555, 0x33333333, 0x0F0F0F0F, 0x00FF00FF, 0x0000FFFF)


def level_intra(Z, g, i, inverse):
    """level i < 5 (r = m - i): pairs of bit positions p, p + 2^i inside each 32-plane value word group."""
    A = LOWMASK[i]
    sh = 1 << i
    planes = [g.plane_off(k) for k in range(32)]
    pmask_bits = sum(1 << b for b in g.plane_bit)
    for w in range(g.nw):
        if w & pmask_bits:
            continue
        W = [int(Z[w | p]) for p in planes]
        if inverse:
            W = [v ^ ((v & A) << sh) for v in W]
        y = [(v >> sh) & A for v in W]
        # per-position masks: M[k][j] bit p = bit k of tau(p) (x) 2^j
        M = [[0] * 32 for _ in range(32)]
        for p in range(32):
            if (p >> i) & 1:
                continue
            rows = mat_rows(point(g.q((w << 5) | p, i)))
            for k in range(32):
                for j in range(32):
                    if (rows[k] >> j) & 1:
                        M[k][j] |= 1 << p
        for k in range(32):
            acc = 0
            for j in range(32):
                acc ^= y[j] & M[k][j]
            W[k] ^= acc
        if not inverse:
            W = [v ^ ((v & A) << sh) for v in W]
        for k in range(32):
            Z[w | planes[k]] = W[k]


def forward(Z, g):
    for r in range(1, g.m + 1):
        if r <= g.m - 5:
            level_word(Z, g, r, False)
        else:
            level_intra(Z, g, g.m - r, False)


def inverse(Z, g):
    for r in range(g.m, 0, -1):
        if r <= g.m - 5:
            level_word(Z, g, r, True)
        else:
            level_intra(Z, g, g.m - r, True)


def values(Z, g):
    """{(word w with plane bits clear, p): 32-bit value}"""
    planes = [g.plane_off(k) for k in range(32)]
    pmask_bits = sum(1 << b for b in g.plane_bit)
    out = {}
    for w in range(g.nw):
        if w & pmask_bits:
            continue
        W = [int(Z[w | p]) for p in planes]
        for p in range(32):
            v = 0
            for k in range(32):
                v |= ((W[k] >> p) & 1) << k
            out[(w, p)] = v
    return out


def pointwise(Z, A, g):
    """Z <- Z (x) A value-wise (both in transformed layout)."""
    planes = [g.plane_off(k) for k in range(32)]
    pmask_bits = sum(1 << b for b in g.plane_bit)
    for w in range(g.nw):
        if w & pmask_bits:
            continue
        Wz = [int(Z[w | p]) for p in planes]
        Wa = [int(A[w | p]) for p in planes]
        out = [0] * 32
        for p in range(32):
            a = 0
            z = 0
            for k in range(32):
                a |= ((Wa[k] >> p) & 1) << k
                z |= ((Wz[k] >> p) & 1) << k
            c = nim_mul(a, z)
            for k in range(32):
                out[k] |= ((c >> k) & 1) << p
        for k in range(32):
            Z[w | planes[k]] = out[k]


# ---- basis conversion on a whole row held as one Python int
def _periodic(mask_fn, chunk, n):
    base = 0
    for l in range(chunk):
        if mask_fn(l):
            base |= 1 << l
    out = 0
    for c in range(n // chunk):
        out |= base << (c * chunk)
    return out


def convert(v, ell, inverse):
    n = 1 << ell
    ph = conversion_phases(ell)
    if inverse:
        ph = list(reversed(ph))
    for chunk, pw, half in ph:
        D = half - pw
        m1 = _periodic(lambda l: (l - half < pw) if l >= half else (l >= pw), chunk, n)
        out = v ^ ((v >> D) & m1)
        if not inverse:
            m2 = _periodic(lambda l: l < half and pw <= l < 2 * pw, chunk, n)
            out ^= (v >> (2 * D)) & m2
        v = out
    return v


def int_to_words(v, nw):
    return np.frombuffer(v.to_bytes(nw * 4, "little"), dtype=np.uint32).copy()


def words_to_int(Z):
    return int.from_bytes(Z.astype("<u4").tobytes(), "little")


def polymul(a, b, m):
    g = Geom(m)
    Za = int_to_words(convert(a, m - 1, False), g.nw)
    Zb = int_to_words(convert(b, m - 1, False), g.nw)
    forward(Za, g)
    forward(Zb, g)
    pointwise(Zb, Za, g)
    inverse(Zb, g)
    return convert(words_to_int(Zb), m, True)


# ---------------------------------------------------------------------------------------------------
# The same pruned transform in "compact" form for any tower field GF(2^F), F = 8, 16, 32 (positions are
# renumbered after every pruned level): small enough to test the math exhaustively on the CPU
# (tests/test_fafft_model.py); checks along the way that tau = 2^r + tau_lo at the pruned levels.
from gf2fft_model import mono_to_novel, novel_to_mono  # noqa: E402


def pruned_levels(F):
    r = 1
    out = []
    while r < F:
        out.append(r); r *= 2
    return out  # relative levels r = m - i that are pruned

def small_fft(c, m, F):
    """c: list of 2^m GF(2) novel-basis coeffs. returns dict-free list of values at cross-section, compact order."""
    PR = set(pruned_levels(F))
    # data: list indexed by compact position; keep track of the list of surviving/unprocessed absolute bits
    data = list(c)
    bits = list(range(m))  # absolute bit i <-> list index position in compact index (ascending)
    for i in range(m - 1, -1, -1):
        r = m - i
        pos = bits.index(i)  # compact bit position of level i
        n = len(data)
        new = None
        if r in PR:
            new = [0] * (n // 2)
        for idx in range(n):
            if (idx >> pos) & 1:
                continue
            # absolute index of higher bits
            jh = 0
            for cp, ab in enumerate(bits):
                if ab > i and (idx >> cp) & 1:
                    jh |= 1 << ab
            tau = point(((1 << m) | jh) >> i)
            x, y = data[idx], data[idx | (1 << pos)]
            if r in PR:
                t = r.bit_length() - 1  # r = 2^t
                assert tau >> r == 1, (hex(tau), r)
                assert x < (1 << r) and y < (1 << r)
                v = x ^ nim_mul(tau, y)
                assert v == (x ^ nim_mul(tau & ((1 << r) - 1), y)) | (y << r)
                # compact: remove bit pos
                lo = idx & ((1 << pos) - 1)
                hi = idx >> (pos + 1)
                new[(hi << pos) | lo] = v
            else:
                x ^= nim_mul(tau, y)
                y ^= x
                data[idx], data[idx | (1 << pos)] = x, y
        if r in PR:
            data = new
            bits.remove(i)
    return data, bits

def small_ifft(vals, m, F):
    PR = set(pruned_levels(F))
    bits = [i for i in range(m) if (m - i) not in PR]
    data = list(vals)
    for i in range(m):
        r = m - i
        if r in PR:
            # insert bit i into bits
            pos = sum(1 for b in bits if b < i)
            n = len(data)
            new = [0] * (2 * n)
            newbits = bits[:pos] + [i] + bits[pos:]
            for idx in range(n):
                lo = idx & ((1 << pos) - 1)
                hi = idx >> pos
                nidx = (hi << (pos + 1)) | lo
                jh = 0
                for cp, ab in enumerate(newbits):
                    if ab > i and (nidx >> cp) & 1:
                        jh |= 1 << ab
                tau = point(((1 << m) | jh) >> i)
                v = data[idx]
                y = v >> r
                x = (v & ((1 << r) - 1)) ^ nim_mul(tau & ((1 << r) - 1), y)
                new[nidx] = x
                new[nidx | (1 << pos)] = y
            data = new
            bits = newbits
        else:
            pos = bits.index(i)
            for idx in range(len(data)):
                if (idx >> pos) & 1:
                    continue
                jh = 0
                for cp, ab in enumerate(bits):
                    if ab > i and (idx >> cp) & 1:
                        jh |= 1 << ab
                tau = point(((1 << m) | jh) >> i)
                x, y = data[idx], data[idx | (1 << pos)]
                y ^= x
                x ^= nim_mul(tau, y)
                data[idx], data[idx | (1 << pos)] = x, y
    return data

def small_polymul(a, b, m, F):
    n = 1 << m
    ca = mono_to_novel([(a >> i) & 1 for i in range(n)])
    cb = mono_to_novel([(b >> i) & 1 for i in range(n)])
    fa, bits = small_fft(ca, m, F)
    fb, _ = small_fft(cb, m, F)
    assert len(fa) == n // F
    fc = [nim_mul(x, y) for x, y in zip(fa, fb)]
    cc = small_ifft(fc, m, F)
    assert all(v in (0, 1) for v in cc), "non-binary coefficients"
    cm = novel_to_mono(cc)
    return sum(v << i for i, v in enumerate(cm))



if __name__ == "__main__":
    import random
    import time

    random.seed(7)
    m = 21
    na = 1 << (m - 1)
    a = random.getrandbits(na)
    b = random.getrandbits(na - 3)
    t0 = time.time()
    c = polymul(a, b, m)
    print("model %.1f s" % (time.time() - t0))
    sys.path.insert(0, "/root/repo")
    from oracle import sot_oracle as orc
    n64 = na // 64
    aw = np.frombuffer(a.to_bytes(n64 * 8, "little"), dtype=np.uint64)
    bw = np.frombuffer(b.to_bytes(n64 * 8, "little"), dtype=np.uint64)
    cw = orc.gf2x_mul(aw, bw)
    want = int.from_bytes(np.asarray(cw, dtype="<u8").tobytes(), "little")
    print("match:", c == want)
