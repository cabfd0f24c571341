"""Executable model of the bit-sliced additive FFT the HIP QC-encode kernels run.

Design aid + table generator (gen_tables.py imports it); the product never
imports it at run time. Pure Python ints, small sizes only.

Field: GF(2^32) as the nimbers below 2^32 (Conway's tower
F_{t+1} = F_t[u_t]/(u_t^2 + u_t + prod_{s<t} u_s), u_t = 2^(2^t)); bit k of a
value is its coordinate on the tower product basis, so multiplying by an element
of the subfield GF(2^(2^t)) acts independently on each aligned group of 2^t bits.

Evaluation points: Cantor basis beta_0 = 1, beta_i^2 + beta_i = beta_{i-1}
(expressed as nimbers). Point p = XOR_i p_i beta_i. With it the subspace
polynomials s_i have GF(2) coefficients, s_i(beta_{i+t}) = beta_t, and the
Lin-Chung-Han butterfly twiddle of block j at level i is point(2j).
"""
from functools import lru_cache

M = 32  # field bits


# ------------------------------------------------------------------ nim product
@lru_cache(maxsize=None)
def _nim_pow2(i, j):
    """(2^i) (x) (2^j)."""
    if i == 0:
        return 1 << j
    if j == 0:
        return 1 << i
    # split off the highest Fermat power present
    t = max(i.bit_length(), j.bit_length()) - 1  # bit index of Fermat exponent 2^t
    F = 1 << t  # exponent of Fermat 2-power u_t = 2^F
    if i & F and j & F:
        # u_t^2 = u_t + u_t/2-ish: (2^F)(x)(2^F) = 2^F xor 2^(F-1)... times the rest
        rest = nim_mul(1 << (i ^ F), 1 << (j ^ F))
        return nim_mul(rest, (1 << F) ^ (1 << (F - 1)))
    if i & F:
        return nim_mul(1 << (i ^ F), 1 << j) << F
    if j & F:
        return nim_mul(1 << i, 1 << (j ^ F)) << F
    raise AssertionError


def nim_mul(a, b):
    r = 0
    i = 0
    x = a
    while x:
        if x & 1:
            y = b
            j = 0
            while y:
                if y & 1:
                    r ^= _nim_pow2(i, j)
                y >>= 1
                j += 1
        x >>= 1
        i += 1
    return r


def nim_sq(a):
    return nim_mul(a, a)


def nim_pow(a, e):
    r = 1
    while e:
        if e & 1:
            r = nim_mul(r, a)
        a = nim_sq(a)
        e >>= 1
    return r


def nim_inv(a):
    return nim_pow(a, (1 << M) - 2)


# ------------------------------------------------------------------ GF(2) linear algebra on ints
def solve_gf2(cols, target, nbits=M):
    """Find x (int bitmask over len(cols) columns) with XOR_{j in x} cols[j] == target, minimal pivots."""
    rows = []  # (value, combo)
    basis = {}
    for j, c in enumerate(cols):
        v, combo = c, 1 << j
        while v:
            hb = v.bit_length() - 1
            if hb in basis:
                bv, bc = basis[hb]
                v ^= bv
                combo ^= bc
            else:
                basis[hb] = (v, combo)
                break
    v, combo = target, 0
    while v:
        hb = v.bit_length() - 1
        if hb not in basis:
            return None
        bv, bc = basis[hb]
        v ^= bv
        combo ^= bc
    return combo


@lru_cache(maxsize=None)
def cantor_basis():
    """beta_0..beta_31 as nimbers: beta_0 = 1, beta_i^2 + beta_i = beta_{i-1}."""
    cols = [nim_sq(1 << j) ^ (1 << j) for j in range(M)]
    beta = [1]
    for i in range(1, M):
        x = solve_gf2(cols, beta[-1])
        assert x is not None
        # two solutions x and x^1 (1 is in the kernel); take the even one for determinism
        x &= ~1
        assert nim_sq(x) ^ x == beta[-1]
        beta.append(x)
    for t in range(6):
        for i in range(min(1 << t, M)):
            assert beta[i] < (1 << (1 << t))
    return tuple(beta)


def point(p):
    """Evaluation point number p: XOR_i p_i beta_i."""
    beta = cantor_basis()
    r = 0
    i = 0
    while p:
        if p & 1:
            r ^= beta[i]
        p >>= 1
        i += 1
    return r


def s_eval(i, x):
    """Subspace polynomial s_i(x) = s_1 applied i times, s_1(y) = y^2 + y."""
    for _ in range(i):
        x = nim_sq(x) ^ x
    return x


# ------------------------------------------------------------------ LCH FFT (scalar model)
def fft(f, ell):
    """f: 2^ell coefficients in the novel basis -> values at point(0..2^ell-1). In place, returns f."""
    n = 1 << ell
    assert len(f) == n
    for i in range(ell - 1, -1, -1):
        half = 1 << i
        for j in range(n >> (i + 1)):
            tw = point(2 * j)
            base = j << (i + 1)
            for k in range(half):
                x, y = f[base + k], f[base + half + k]
                x ^= nim_mul(tw, y)
                y ^= x
                f[base + k], f[base + half + k] = x, y
    return f


def ifft(f, ell):
    n = 1 << ell
    for i in range(ell):
        half = 1 << i
        for j in range(n >> (i + 1)):
            tw = point(2 * j)
            base = j << (i + 1)
            for k in range(half):
                x, y = f[base + k], f[base + half + k]
                y ^= x
                x ^= nim_mul(tw, y)
                f[base + k], f[base + half + k] = x, y
    return f


def novel_eval(f, x):
    """Evaluate sum f_k X_k(x), X_k = prod s_i(x)^{k_i}, directly."""
    ell = (len(f) - 1).bit_length()
    sv = [s_eval(i, x) for i in range(max(ell, 1))]
    r = 0
    for k, c in enumerate(f):
        if c:
            t = c
            i = 0
            kk = k
            while kk:
                if kk & 1:
                    t = nim_mul(t, sv[i])
                kk >>= 1
                i += 1
            r ^= t
    return r


# ------------------------------------------------------------------ basis conversion: phase list
def conversion_phases(ell):
    """Monomial -> novel basis conversion on 2^ell coefficients as an ordered list of phases
    (chunk, pw, half), each meaning, for every aligned chunk of `chunk` coefficients and local
    index l (D = half - pw):
        out[i] = in[i] ^ m1(l) in[i + D] ^ m2(l) in[i + 2D]
        m1(l) = (l >= half) ? (l - half < pw) : (l >= pw)
        m2(l) = (l < half) and (pw <= l < 2 pw)
    This is the Taylor expansion at y = x^q + x (q = 2^(2^t)) used recursively on the bit fields
    of the index; derived from the Cantor-basis identity s_{i+2^t} = s_i o s_{2^t}.
    The inverse conversion runs the list backwards with m2 = 0.
    """
    phases = []

    def conv(nbits, unit):
        # convert a polynomial with 2^nbits coefficients, each coefficient a unit of `unit` entries
        if nbits <= 1:
            return
        t = 0
        while (1 << (t + 1)) < nbits:
            t += 1
        qb = 1 << t  # q = 2^qb coefficients per block
        R = nbits - qb  # 2^R blocks
        # Taylor expansion at x^q + x: pow = 2^(R-1) .. 1
        for pb in range(R - 1, -1, -1):
            pw = (1 << pb) * unit
            half = (1 << (pb + qb)) * unit
            phases.append((2 * half, pw, half))
        conv(R, unit << qb)  # y direction (blocks as units)
        conv(qb, unit)  # x direction inside each block

    conv(ell, 1)
    return phases


def apply_phase(vec, phase, inverse=False):
    chunk, pw, half = phase
    D = half - pw
    n = len(vec)
    out = list(vec)
    for i in range(n):
        l = i % chunk
        if l >= half:
            m1 = (l - half) < pw
            m2 = False
        else:
            m1 = l >= pw
            m2 = (pw <= l < 2 * pw) and not inverse
        if m1:
            out[i] ^= vec[i + D]
        if m2:
            out[i] ^= vec[i + 2 * D]
    return out


def mono_to_novel(f):
    ell = (len(f) - 1).bit_length()
    for ph in conversion_phases(ell):
        f = apply_phase(f, ph)
    return f


def novel_to_mono(f):
    ell = (len(f) - 1).bit_length()
    for ph in reversed(conversion_phases(ell)):
        f = apply_phase(f, ph, inverse=True)
    return f


def mono_eval(f, x):
    r = 0
    for c in reversed(f):
        r = nim_mul(r, x) ^ c
    return r


# ------------------------------------------------------------------ Kronecker substitution x -> g
W = 16  # polynomial bits packed per field element


@lru_cache(maxsize=None)
def kron():
    """Returns (g, G, Ginv): G[t] = g^t as nimber (t < 32); Ginv[k] = bitmask of poly coefficients
    (over powers of g) of the nimber 2^k."""
    g = None
    best = None
    for cand in range(1 << 16, (1 << 16) + 4096):
        pw = [1]
        for _ in range(31):
            pw.append(nim_mul(pw[-1], cand))
        # need independence of 1..g^31
        inv = [solve_gf2(pw, 1 << k) for k in range(M)]
        if any(v is None for v in inv):
            continue
        weight = sum(bin(v).count("1") for v in inv) + sum(bin(v).count("1") for v in pw[:16])
        if best is None or weight < best[0]:
            best = (weight, cand, tuple(pw), tuple(inv))
        if cand > (1 << 16) + 64 and best is not None:
            break
    _, g, G, Ginv = best
    return g, G, Ginv


def pack_elem(bits16):
    _, G, _ = kron()
    r = 0
    for t in range(W):
        if (bits16 >> t) & 1:
            r ^= G[t]
    return r


def unpack_elem(v):
    _, _, Ginv = kron()
    r = 0
    k = 0
    while v:
        if v & 1:
            r ^= Ginv[k]
        v >>= 1
        k += 1
    return r


# ------------------------------------------------------------------ full product model
def polymul_bits(a, b, nbits, ell):
    """a, b: Python ints holding GF(2)[x] polynomials of < nbits bits, nbits <= 16 * 2^(ell-1).
    Returns a*b through pack -> convert -> FFT -> pointwise -> IFFT -> convert back -> unpack."""
    L = 1 << ell
    half = L // 2
    assert nbits <= W * half

    def to_elems(v):
        return [pack_elem((v >> (W * e)) & 0xFFFF) for e in range(half)] + [0] * half

    fa = fft(mono_to_novel(to_elems(a)), ell)
    fb = fft(mono_to_novel(to_elems(b)), ell)
    fc = [nim_mul(x, y) for x, y in zip(fa, fb)]
    c = novel_to_mono(ifft(fc, ell))
    r = 0
    for e, v in enumerate(c):
        r ^= unpack_elem(v) << (W * e)
    return r


def clmul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a <<= 1
        b >>= 1
    return r


if __name__ == "__main__":
    import random

    random.seed(1)
    # field sanity
    for _ in range(50):
        a, b, c = (random.getrandbits(32) for _ in range(3))
        assert nim_mul(a, nim_mul(b, c)) == nim_mul(nim_mul(a, b), c)
        assert nim_mul(a, b ^ c) == nim_mul(a, b) ^ nim_mul(a, c)
        if a:
            assert nim_mul(a, nim_inv(a)) == 1
    beta = cantor_basis()
    print("beta:", [hex(x) for x in beta[:8]], "...")
    # subfield block structure: c in GF(2^8), x arbitrary
    for _ in range(20):
        c8 = random.getrandbits(8)
        x = random.getrandbits(32)
        want = 0
        for h in range(4):
            want |= nim_mul(c8, (x >> (8 * h)) & 0xFF) << (8 * h)
        assert nim_mul(c8, x) == want
    # FFT == direct evaluation of the novel-basis polynomial
    for ell in (1, 2, 3, 5):
        f = [random.getrandbits(32) for _ in range(1 << ell)]
        vals = fft(list(f), ell)
        for p in range(1 << ell):
            assert vals[p] == novel_eval(f, point(p)), (ell, p)
        assert ifft(list(vals), ell) == f
    print("fft ok")
    # conversion: monomial poly evaluated directly == FFT(mono_to_novel)
    for ell in (1, 2, 3, 4, 5, 6, 7):
        f = [random.getrandbits(32) for _ in range(1 << ell)]
        g = mono_to_novel(list(f))
        for p in random.sample(range(1 << ell), min(8, 1 << ell)):
            assert mono_eval(f, point(p)) == novel_eval(g, point(p)), (ell, p)
        assert novel_to_mono(list(g)) == f
    print("conversion ok")
    for ell, nbits in ((3, 64), (5, 256), (6, 500), (8, 2048)):
        a = random.getrandbits(nbits)
        b = random.getrandbits(nbits)
        assert polymul_bits(a, b, nbits, ell) == clmul(a, b), ell
    print("polymul ok; g = %#x" % kron()[0])
