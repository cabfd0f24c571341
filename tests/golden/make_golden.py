"""Generates tests/golden/silent_ot_n1000.{npz,json} (n = 1000, T = 2), silent_ot_n65536.{npz,json} (n = 65 536, T = 4)
and gf2x_products.json (three raw polynomial products) from the CPU oracle. The reference is Rust and cannot run here, so these vectors pin the ORACLE (itself pinned to
the reference's KATs / properties / bitpolymul2, see tests/test_oracle.py) and every later HIP build against
regressions; they are data (inputs + expected digests), produced by this script:

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import sot_oracle as orc  # noqa: E402


def dig(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pipeline_fixture(name, n, T, rng_seed):
    conf = orc.configure(n)
    rng = np.random.default_rng(rng_seed)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    smsgs, B, rT_s, sums, last = orc.random_silent_send(conf, msg2, delta, seed, T)
    cm_s = orc.last_cmodp()
    rmsgs, ch, A, rT_r = orc.random_silent_receive(conf, msg, choices, sums, last)
    a, b, c = orc.mt_bits(smsgs.reshape(-1, 2), rmsgs, ch)
    # ChoiceBitPacking::False on the same receiver leaves (quasi_cyclic_encode.rs:104-117)
    A_un, C_un = orc.qc_dual_encode2(conf, rT_r, orc.pprf_points(conf, choices), packed=False)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), msg2=msg2, msg=msg, choices=choices, delta=delta,
                        seed=seed, B_head=B[:64], A_head=A[:64], B_tail=B[-64:], A_tail=A[-64:], smsgs_head=smsgs[:32],
                        rmsgs_head=rmsgs[:64], choice_words=ch, mt_a=a, mt_b=b, mt_c=c, tg_last=last)
    meta = {
        "n": n, "T": T, "conf": conf.as_dict(),
        "sha256": {
            "sender_rT": dig(rT_s), "receiver_rT": dig(rT_r), "tg_sums": dig(sums), "tg_last": dig(last),
            "a_poly": dig(orc.qc_a_poly(conf)), "cmodp_row0": dig(cm_s[0]), "cmodp_row1": dig(cm_s[1]),
            "cmodp_row127": dig(cm_s[127]), "B": dig(B), "A": dig(A), "sender_msgs": dig(smsgs),
            "receiver_msgs": dig(rmsgs), "choice_words": dig(ch), "mt_a": dig(a), "mt_b": dig(b), "mt_c": dig(c),
            "A_unpacked": dig(A_un), "C_unpacked": dig(C_un),
        },
    }
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(name, "written:", meta["sha256"]["B"][:16])


PRODUCT_N64 = (264, 16386, 262146)  # n64 of n = 1000, 2^20, 2^24


def product_operands(n64):
    rng = np.random.default_rng(1000003 + n64)
    return rng.integers(0, 2 ** 64, n64, dtype=np.uint64), rng.integers(0, 2 ** 64, n64, dtype=np.uint64)


def product_fixture():
    """Raw GF(2)[x] products a * b of n64-word operands (2 n64 result words, low degree first): operands are regenerated
    from a seed, the fixture holds the digest and the first / last 8 words of the product. Computed with the oracle's
    own multiplier and, when oracle/_ref exists, checked against the reference's bitpolymul_2_128 before writing."""
    out = {}
    for n64 in PRODUCT_N64:
        a, b = product_operands(n64)
        c = orc.gf2x_mul(a, b)
        if orc.ref_lib() is not None:
            assert np.array_equal(c, orc.ref_bitpolymul_2_128(a, b)), n64
        out[str(n64)] = {"sha256": dig(c), "head": [int(x) for x in c[:8]], "tail": [int(x) for x in c[-8:]],
                         "checked_against_reference_bitpolymul_2_128": orc.ref_lib() is not None}
    with open(os.path.join(HERE, "gf2x_products.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("products written")


def main():
    pipeline_fixture("silent_ot_n1000", 1000, 2, 20261110)
    pipeline_fixture("silent_ot_n65536", 65536, 4, 20261111)
    product_fixture()


if __name__ == "__main__":
    main()
