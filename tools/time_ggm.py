#!/usr/bin/env python3
"""per-kernel device time of the GGM stage and the hash stage at n = 2^log2n (HIP events, sot_prof_*); FLUTE_SOT_LIB picks the library.
usage: time_ggm.py [log2n] [reps]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import flute_amd as fa

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
eng = fa.Engine(1 << log2n)
rng = np.random.default_rng(1)
x = fa.fake_base(eng.conf, rng)
d, s = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(2))
tg = eng.sender_expand(x[0], d, s, 4)
eng.receiver_expand(x[1], x[2], *tg)
eng.prof_reset()
eng.prof_enable(True)
for _ in range(reps):
    tg = eng.sender_expand(x[0], d, s, 4)
    eng.receiver_expand(x[1], x[2], *tg)
eng.prof_enable(False)
p = eng.prof_read()
tot = 0.0
for k, (ms, cnt) in sorted(p.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %8.4f ms/step  %5d launches/step" % (k, ms / reps, cnt // reps))
    tot += ms / reps
print("total %.4f ms per step (sender + receiver expansion)" % tot)
