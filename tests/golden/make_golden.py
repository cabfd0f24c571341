"""Generates tests/golden/silent_ot_n1000.npz + .json from the CPU oracle (n = 1000, T = 2 and a 40-tree
PPRF-only case). The reference is Rust and cannot run here, so these vectors pin the ORACLE (itself pinned to
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


def main():
    n, T = 1000, 2
    conf = orc.configure(n)
    rng = np.random.default_rng(20261110)
    msg2, msg, choices = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    smsgs, B, rT_s, sums, last = orc.random_silent_send(conf, msg2, delta, seed, T)
    cm_s = orc.last_cmodp()
    rmsgs, ch, A, rT_r = orc.random_silent_receive(conf, msg, choices, sums, last)
    a, b, c = orc.mt_bits(smsgs.reshape(-1, 2), rmsgs, ch)
    np.savez_compressed(os.path.join(HERE, "silent_ot_n1000.npz"), msg2=msg2, msg=msg, choices=choices, delta=delta,
                        seed=seed, B_head=B[:64], A_head=A[:64], smsgs_head=smsgs[:32], rmsgs_head=rmsgs[:64],
                        choice_words=ch, mt_a=a, mt_b=b, mt_c=c, tg_last=last)
    meta = {
        "n": n, "T": T, "conf": conf.as_dict(),
        "sha256": {
            "sender_rT": dig(rT_s), "receiver_rT": dig(rT_r), "tg_sums": dig(sums), "tg_last": dig(last),
            "a_poly": dig(orc.qc_a_poly(conf)), "cmodp_row0": dig(cm_s[0]), "cmodp_row1": dig(cm_s[1]),
            "cmodp_row127": dig(cm_s[127]), "B": dig(B), "A": dig(A), "sender_msgs": dig(smsgs),
            "receiver_msgs": dig(rmsgs), "choice_words": dig(ch), "mt_a": dig(a), "mt_b": dig(b), "mt_c": dig(c),
        },
    }
    with open(os.path.join(HERE, "silent_ot_n1000.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("golden written:", meta["sha256"]["B"][:16])


if __name__ == "__main__":
    main()
