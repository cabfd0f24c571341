"""The identity behind the Karatsuba application of the twiddle matrices (csrc/qc.hip: kara_acc, kara_lds,
k_build_lowmasks, k_build_amat): in the nimber tower GF(2^N) = GF(2^(N/2))[t], t^2 = t + zeta, the
multiplication matrix of any element is [[A, X], [B, A + B]] in half-size blocks, and every block is again the
matrix of an element of the subfield. Checked on the same tables the kernels are built from (csrc/fafft_model.mat_rows)."""
import os
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flute_amd", "csrc"))
from fafft_model import mat_rows  # noqa: E402
from gf2fft_model import nim_mul  # noqa: E402


def blocks(rows, k0, j0, n):
    """n x n block at rows k0.., columns j0.. as a list of row masks"""
    return [(rows[k0 + k] >> j0) & ((1 << n) - 1) for k in range(n)]


def check(rows, k0, j0, n):
    if n == 1:
        return
    h = n // 2
    a, x = blocks(rows, k0, j0, h), blocks(rows, k0, j0 + h, h)
    b, ab = blocks(rows, k0 + h, j0, h), blocks(rows, k0 + h, j0 + h, h)
    assert ab == [p ^ q for p, q in zip(a, b)]
    for (kk, jj) in ((k0, j0), (k0, j0 + h), (k0 + h, j0 + h)):  # the three blocks the kernels read
        check(rows, kk, jj, h)


def test_block_structure_of_multiplication_matrices():
    rng = random.Random(7)
    for _ in range(100):
        check(mat_rows(rng.getrandbits(32)), 0, 0, 32)
    for _ in range(50):
        check(mat_rows(rng.getrandbits(16)), 0, 0, 16)


def test_karatsuba_product_equals_direct_product():
    """out_lo = A y_lo + X y_hi, out_hi = (A + B)(y_lo + y_hi) + A y_lo, applied recursively down to 2 x 2, against nim_mul"""
    rng = random.Random(11)

    def apply(rows, k0, j0, n, y):
        """(block at k0, j0) * y, y an n-bit vector"""
        if n <= 2:
            out = 0
            for k in range(n):
                if bin(((rows[k0 + k] >> j0) & ((1 << n) - 1)) & y).count("1") & 1:
                    out |= 1 << k
            return out
        h = n // 2
        ylo, yhi = y & ((1 << h) - 1), y >> h
        t1 = apply(rows, k0, j0, h, ylo)
        lo = t1 ^ apply(rows, k0, j0 + h, h, yhi)
        hi = t1 ^ apply(rows, k0 + h, j0 + h, h, ylo ^ yhi)
        return lo | (hi << h)

    for _ in range(200):
        c, y = rng.getrandbits(32), rng.getrandbits(32)
        assert apply(mat_rows(c), 0, 0, 32, y) == nim_mul(c, y)
