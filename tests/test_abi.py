"""CPU-side checks of the C-ABI library: it loads and exports every symbol include/flute_sot.h declares,
and the pure-host entry points (configuration, point arithmetic) agree with the oracle. No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "flute_sot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sot_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from flute_amd.lib import EXPORTS, load_library
    L = load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(EXPORTS) == declared


@pytest.mark.parametrize("n", [1000, 65536, 2000000, 1 << 20, 1 << 24, 1 << 26])
def test_sot_configure_matches_oracle(orc, n):
    from flute_amd import QuasiCyclicConf
    got = QuasiCyclicConf.configure(n).c.as_dict()
    assert got == orc.configure(n).as_dict()


def test_points_match_oracle(orc):
    from flute_amd import QuasiCyclicConf, sample_base_choice_bits
    from flute_amd.lib import load_library
    conf = QuasiCyclicConf.configure(1000)
    rng = np.random.default_rng(5)
    ch = sample_base_choice_bits(conf, rng)
    pts = np.zeros(conf.parts, dtype=np.uint64)
    rc = load_library().sot_receiver_points(C.byref(conf.c), ch.ctypes.data_as(C.c_void_p),
                                            pts.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert np.array_equal(pts, orc.pprf_points(orc.configure(1000), ch))
    assert (pts < conf.N2).all()


def test_bad_arguments_return_errors_not_aborts():
    from flute_amd.lib import Conf, load_library
    L = load_library()
    c = Conf()
    assert L.sot_configure(0, 2, 128, C.byref(c)) != 0
    assert L.sot_configure(1000, 1, 128, C.byref(c)) != 0
    assert b"scaler" in L.sot_last_error()
    h = C.c_void_p()
    assert L.sot_ctx_create(1000, 3, 0, C.byref(h)) != 0  # only scaler 2
    assert L.sot_sender_hash(None, None, None) != 0


def test_mul_triples_split_off_last():
    from flute_amd.mul_triple import MulTriples
    rng = np.random.default_rng(1)
    n = 200
    bits = [rng.integers(0, 2, n, dtype=np.uint8) for _ in range(3)]
    mts = MulTriples(*[MulTriples._pack(b) for b in bits], n)
    last = mts.split_off_last(70)
    assert len(last) == 70 and len(mts) == 130
    for got, b in zip((last.a, last.b, last.c), bits):
        assert np.array_equal(MulTriples._bits(got, 70), b[130:])
    for got, b in zip((mts.a, mts.b, mts.c), bits):
        assert np.array_equal(MulTriples._bits(got, 130), b[:130])


def test_sys_crate_binds_every_declared_symbol():
    """crates/flute-sot-sys/src/lib.rs (the Rust `-sys` binding a FLUTE maintainer adds) names
    exactly the symbols of include/flute_sot.h, and its SotConf mirrors sot_conf field for field."""
    rs = open(os.path.join(ROOT, "crates", "flute-sot-sys", "src", "lib.rs")).read()
    bound = sorted(set(re.findall(r"pub fn (sot_[A-Za-z0-9_]+)\(", rs)))
    assert bound == _declared_symbols()
    from flute_amd.lib import Conf, ShardInfo
    for struct, cls in (("SotConf", Conf), ("SotShardInfo", ShardInfo)):
        body = re.search(r"pub struct %s \{(.*?)\n\}" % struct, rs, flags=re.S).group(1)
        fields = re.findall(r"pub ([A-Za-z0-9_]+): (u32|u64)", body)
        want = [(k, "u64" if C.sizeof(t) == 8 else "u32") for k, t in cls._fields_]
        assert fields == want
