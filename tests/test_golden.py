"""Golden fixtures (tests/golden, made by make_golden.py from the oracle): the oracle must keep reproducing
them on CPU, and the HIP path must reproduce them on the GPU, digest for digest."""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def dig(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module", params=["silent_ot_n1000", "silent_ot_n65536"])
def gold(request):
    meta = json.load(open(os.path.join(HERE, "golden", request.param + ".json")))
    data = np.load(os.path.join(HERE, "golden", request.param + ".npz"), allow_pickle=False)
    return meta, data


def test_oracle_reproduces_golden(orc, gold):
    meta, d = gold
    conf = orc.configure(meta["n"])
    assert conf.as_dict() == meta["conf"]
    smsgs, B, rT_s, sums, last = orc.random_silent_send(conf, d["msg2"], d["delta"], d["seed"], meta["T"])
    cm = orc.last_cmodp()
    rmsgs, ch, A, rT_r = orc.random_silent_receive(conf, d["msg"], d["choices"], sums, last)
    a, b, c = orc.mt_bits(smsgs.reshape(-1, 2), rmsgs, ch)
    got = {"sender_rT": dig(rT_s), "receiver_rT": dig(rT_r), "tg_sums": dig(sums), "tg_last": dig(last),
           "a_poly": dig(orc.qc_a_poly(conf)), "cmodp_row0": dig(cm[0]), "cmodp_row1": dig(cm[1]),
           "cmodp_row127": dig(cm[127]), "B": dig(B), "A": dig(A), "sender_msgs": dig(smsgs),
           "receiver_msgs": dig(rmsgs), "choice_words": dig(ch), "mt_a": dig(a), "mt_b": dig(b), "mt_c": dig(c)}
    A_un, C_un = orc.qc_dual_encode2(conf, rT_r, orc.pprf_points(conf, d["choices"]), packed=False)
    got["A_unpacked"], got["C_unpacked"] = dig(A_un), dig(C_un)
    assert got == meta["sha256"]
    assert np.array_equal(B[:64], d["B_head"]) and np.array_equal(A[:64], d["A_head"])
    assert np.array_equal(B[-64:], d["B_tail"]) and np.array_equal(A[-64:], d["A_tail"])


def test_product_vectors(orc):
    """three raw GF(2)[x] products (n64 = 264, 16 386, 262 146: the row lengths of n = 1000, 2^20, 2^24) against the
    committed digests; with oracle/_ref present the reference's own bitpolymul_2_128 must give the same words"""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import product_operands
    want = json.load(open(os.path.join(HERE, "golden", "gf2x_products.json")))
    for key, w in want.items():
        a, b = product_operands(int(key))
        c = orc.gf2x_mul(a, b)
        assert dig(c) == w["sha256"], key
        assert [int(x) for x in c[:8]] == w["head"] and [int(x) for x in c[-8:]] == w["tail"]
        if orc.ref_lib() is not None:
            assert dig(orc.ref_bitpolymul_2_128(a, b)) == w["sha256"], "reference bitpolymul_2_128 at n64=" + key


@pytest.mark.gpu
def test_gpu_reproduces_golden(gold):
    import flute_amd as fa
    meta, d = gold
    eng = fa.Engine(meta["n"])
    try:
        sums, last = eng.sender_expand(d["msg2"], d["delta"], d["seed"], meta["T"])
        want = meta["sha256"]
        assert dig(sums) == want["tg_sums"] and dig(last) == want["tg_last"]
        assert dig(eng.download_rT(0)) == want["sender_rT"]
        eng.qc_encode(0)
        assert dig(eng.download_encoded(0)) == want["B"]
        smsgs = eng.sender_hash(d["delta"])
        assert dig(smsgs) == want["sender_msgs"]
        eng.receiver_expand(d["msg"], d["choices"], sums, last)
        assert dig(eng.download_rT(1)) == want["receiver_rT"]
        eng.qc_encode(1, eng.points(d["choices"]))
        assert dig(eng.download_encoded(1)) == want["A"]
        rmsgs, ch = eng.receiver_hash()
        assert dig(rmsgs) == want["receiver_msgs"] and dig(ch) == want["choice_words"]
        a, b, c = eng.mt_bits(d["delta"])
        assert (dig(a), dig(b), dig(c)) == (want["mt_a"], want["mt_b"], want["mt_c"])
        Cv = eng.qc_encode2(eng.points(d["choices"]), packed=False)
        assert dig(eng.download_encoded(1)) == want["A_unpacked"] and dig(Cv) == want["C_unpacked"]
    finally:
        eng.close()
