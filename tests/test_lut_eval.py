"""Oracle of the FLUTE online LUT evaluation (oracle/lut_eval_oracle.py): the restated compute_delta_share
reconstructs the plaintext LUT output for every input; fixture sanity (sbox8.lut = FIPS-197 S-box). CPU only."""
import json
import os

import numpy as np
import pytest

from oracle import lut_eval_oracle as le

HERE = os.path.dirname(os.path.abspath(__file__))


def _sbox8():
    d = json.load(open(os.path.join(HERE, "golden", "sbox8_lut.json")))
    return [le.table_bits(l["table_hex"], 8) for l in d["luts"]]


def _aes_sbox():
    def mul(a, b):
        r = 0
        while b:
            if b & 1:
                r ^= a
            a <<= 1
            if a & 0x100:
                a ^= 0x11b
            b >>= 1
        return r
    inv = [0] * 256
    for a in range(1, 256):
        for b in range(1, 256):
            if mul(a, b) == 1:
                inv[a] = b
                break
    S = []
    for x in range(256):
        y, r = inv[x], 0
        for i in range(8):
            bit = ((y >> i) ^ (y >> ((i + 4) % 8)) ^ (y >> ((i + 5) % 8)) ^ (y >> ((i + 6) % 8)) ^ (y >> ((i + 7) % 8)) ^
                   (0x63 >> i)) & 1
            r |= bit << i
        S.append(r)
    return S


def test_sbox8_fixture_is_the_fips197_sbox():
    """bit i of the table NUMBER (Lsb0) of dout[k] = bit k of S[i]; under the reference's parser (Msb0 index, first wire =
    MSB of the row) the circuit therefore maps a row index idx to S[255 - idx]"""
    tabs = _sbox8()
    S = _aes_sbox()
    assert S[0] == 0x63 and S[0x53] == 0xed
    for idx in range(256):
        y = sum(tabs[k][idx] << k for k in range(8))
        assert y == S[255 - idx]


@pytest.mark.parametrize("p", [1, 2, 3, 5])
def test_delta_shares_reconstruct_the_plaintext_small(p):
    rng = np.random.default_rng(p)
    for trial in range(6):
        table = [int(b) for b in rng.integers(0, 2, 1 << p)]
        if trial == 0:
            table = [0] * (1 << p)
            table[-1] = 1  # AND of all inputs (LutGate::and for p = 2)
        for x in range(1 << p):
            xin = [(x >> (p - 1 - i)) & 1 for i in range(p)]
            lam0 = [int(b) for b in rng.integers(0, 2, p)]
            lam1 = [int(b) for b in rng.integers(0, 2, p)]
            lam = [a ^ b for a, b in zip(lam0, lam1)]
            m = [a ^ b for a, b in zip(xin, lam)]
            prods = le.lambda_products(lam)
            e0 = [int(b) for b in rng.integers(0, 2, len(prods))]
            e1 = [a ^ b for a, b in zip(prods, e0)]
            o0, o1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            d0 = le.compute_delta_share(0, table, m, lam0, e0, o0)
            d1 = le.compute_delta_share(1, table, m, lam1, e1, o1)
            assert (d0 ^ d1) ^ (o0 ^ o1) == le.plain_eval(table, xin)


def test_delta_shares_reconstruct_the_sbox():
    tabs = _sbox8()
    rng = np.random.default_rng(8)
    for x in rng.integers(0, 256, 6):
        xin = [(int(x) >> (7 - i)) & 1 for i in range(8)]
        lam0 = [int(b) for b in rng.integers(0, 2, 8)]
        lam1 = [int(b) for b in rng.integers(0, 2, 8)]
        lam = [a ^ b for a, b in zip(lam0, lam1)]
        m = [a ^ b for a, b in zip(xin, lam)]
        prods = le.lambda_products(lam)
        e0 = [int(b) for b in rng.integers(0, 2, len(prods))]
        e1 = [a ^ b for a, b in zip(prods, e0)]
        for k in (0, 5):
            o0, o1 = int(rng.integers(0, 2)), int(rng.integers(0, 2))
            d0 = le.compute_delta_share(0, tabs[k], m, lam0, e0, o0)
            d1 = le.compute_delta_share(1, tabs[k], m, lam1, e1, o1)
            assert (d0 ^ d1) ^ (o0 ^ o1) == le.plain_eval(tabs[k], xin)


def test_aes128_lut_circuit_walk_is_fips197():
    """flute_amd/aes_lut.py: the XOR-linear wiring around the 200 S-boxes, with a table S-box in place of the LUT protocol:
    FIPS-197 appendix C.1; and the lambda walk (no constants) is the linear part of the same map"""
    from flute_amd import aes_lut as al
    S = np.array(_aes_sbox(), dtype=np.uint8)
    key = np.arange(16, dtype=np.uint8)[None]
    pt = np.array([0x11 * i for i in range(16)], dtype=np.uint8)[None]
    calls = []

    def sbox(rnd, x):
        calls.append((rnd, x.shape))
        return al.bytes_to_bits(S[al.bits_to_bytes(x)])
    ct = al.bits_to_bytes(al.aes128_walk(al.bytes_to_bits(pt), al.bytes_to_bits(key), sbox, True))
    assert ct.tobytes().hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"
    assert [c[0] for c in calls] == list(range(10)) and all(c[1] == (1, 20, 8) for c in calls)
    assert 10 * 20 == 200  # AES128_SBOXES of flute_amd/lut_setup.py
