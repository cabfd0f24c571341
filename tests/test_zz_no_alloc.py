"""The one entry point that allocates nothing, sot_alsz_receive_noalloc (include/flute_sot.h): called through the C ABI
and compared bit for bit with the existing sot_alsz_receive on the same inputs AND with the CPU oracle
(oracle/alsz_oracle.py), on two calls in a row, the second with a smaller input (the shared scratch array still holds the
first call's bytes). Collected last on purpose."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _both(L, base, choice, count):
    u0 = np.zeros((128, count // 8), dtype=np.uint8)
    m0 = np.zeros((count, 2), dtype=np.uint64)
    u1 = np.full((128, count // 8), 0xA5, dtype=np.uint8)
    m1 = np.full((count, 2), 0x5A5A5A5A5A5A5A5A, dtype=np.uint64)
    assert L.sot_alsz_receive(0, _p(base), _p(choice), count, _p(u0), _p(m0)) == 0, L.sot_last_error()
    assert L.sot_alsz_receive_noalloc(0, _p(base), _p(choice), count, _p(u1), _p(m1)) == 0, L.sot_last_error()
    assert np.array_equal(u0, u1), "u rows differ at count=%d" % count
    assert np.array_equal(m0, m1), "messages differ at count=%d" % count
    from oracle import alsz_oracle as az
    ou, om = az.receive_random(base, choice.tobytes(), count)
    assert np.array_equal(u1, ou), "u rows differ from the oracle at count=%d" % count
    assert np.array_equal(m1, om), "messages differ from the oracle at count=%d" % count
    return m1


def test_noalloc_entry_point_matches_alsz_receive():
    import flute_amd
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device visible: GPU tests need an MI355X")
    L = flute_amd.load_library()
    cap = int(L.sot_alsz_noalloc_max_count())
    assert cap == 1 << 16
    rng = np.random.default_rng(77)
    base = rng.integers(0, 2 ** 64, size=(128, 2, 2), dtype=np.uint64)
    # first call: the largest count the scratch array takes; second call: smaller and ragged (not a multiple of 128),
    # with other choices, so stale padding or stale rows of the first call would show
    ch1 = rng.integers(0, 256, cap // 8, dtype=np.uint8)
    first = _both(L, base, ch1, cap)
    small = 8 * 1001
    ch2 = rng.integers(0, 256, small // 8, dtype=np.uint8)
    base2 = rng.integers(0, 2 ** 64, size=(128, 2, 2), dtype=np.uint64)
    second = _both(L, base2, ch2, small)
    assert not np.array_equal(first[:small], second)  # other base OTs: other messages
    # beyond the maximum: a clear error, nothing computed
    big = cap + 8
    chb = np.zeros(big // 8, dtype=np.uint8)
    ub = np.zeros((128, big // 8), dtype=np.uint8)
    mb = np.zeros((big, 2), dtype=np.uint64)
    assert L.sot_alsz_receive_noalloc(0, _p(base), _p(chb), big, _p(ub), _p(mb)) != 0
    assert b"exceeds" in L.sot_last_error()
    assert not mb.any()
    assert L.sot_alsz_receive_noalloc(0, _p(base), _p(chb), 12, _p(ub), _p(mb)) != 0  # not a multiple of 8
