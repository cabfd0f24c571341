"""The math of the Frobenius-pruned additive FFT (flute_amd/csrc/fafft_model.py) against schoolbook products, in
the small tower fields where it can be run exhaustively on the CPU. The in-place m = 21 model over GF(2^32), whose
layout the HIP kernels follow word for word, takes minutes in pure Python: it was checked once against the oracle's
GF(2)[x] product (python flute_amd/csrc/fafft_model.py) and its stage outputs are the fixtures of
tests/test_gpu_fafft.py."""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "flute_amd", "csrc"))


@pytest.mark.parametrize("F,m", [(8, 4), (8, 5), (8, 7), (16, 8), (16, 9), (16, 11)])
def test_pruned_transform_multiplies(F, m):
    import fafft_model as fm
    from gf2fft_model import clmul
    rnd = random.Random(F * 100 + m)
    for _ in range(2):
        na = rnd.randrange(1, 1 << m)
        a = rnd.getrandbits(na)
        b = rnd.getrandbits((1 << m) - na)
        assert fm.small_polymul(a, b, m, F) == clmul(a, b)


def test_cross_section_is_one_point_per_frobenius_orbit():
    # index map of squaring on beta_m + W_m is j -> j ^ (j >> 1); the kept indices (pruned bits zero) must hit
    # every orbit exactly once
    m, F = 9, 16
    pruned = [m - r for r in (1, 2, 4, 8)]
    keep = [j for j in range(1 << m) if all(not (j >> b) & 1 for b in pruned)]
    seen = set()
    for j in keep:
        full = j | (1 << m)
        orbit = set()
        x = full
        for _ in range(F):
            orbit.add(x)
            x ^= x >> 1
        assert x == full and len(orbit) == F
        assert not (orbit & seen)
        seen |= orbit
    assert len(seen) == 1 << m
