"""Manual check: GPU pipeline vs oracle (reference multiplier) at arbitrary n. usage: tools/check_n.py n [T]"""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time
import numpy as np
import flute_amd as fa
from oracle import sot_oracle as orc
n = int(eval(sys.argv[1])); T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
conf = orc.configure(n)
print("n", n, conf.as_dict(), flush=True)
rng = np.random.default_rng(n % 1000003)
msg2, msg, choices = orc.fake_base(conf, rng)
delta = rng.integers(0, 2**64, 2, dtype=np.uint64); seed = rng.integers(0, 2**64, 2, dtype=np.uint64)
t0 = time.time(); eng = fa.Engine(n); print("engine %.2fs, %.2f GiB" % (time.time()-t0, eng.device_bytes()/2**30), flush=True)
t0 = time.time()
sums, last = eng.sender_expand(msg2, delta, seed, T); eng.qc_encode(0)
eng.receiver_expand(msg, choices, sums, last); pts = eng.points(choices); eng.qc_encode(1, pts)
a, b, c = eng.mt_bits(delta)
print("gpu %.3fs" % (time.time()-t0), flush=True)
B = eng.download_encoded(0); A = eng.download_encoded(1)
orc.use_reference_multiplier(True)
t0 = time.time()
o_rT, o_sums, o_last = orc.pprf_sender_expand(conf, msg2, delta, seed, T)
print("oracle ggm %.1fs" % (time.time()-t0), flush=True)
assert np.array_equal(sums, o_sums) and np.array_equal(last, o_last)
o_B = orc.qc_dual_encode(conf, o_rT)
print("oracle encode %.1fs" % (time.time()-t0), flush=True)
print("sender B equal:", np.array_equal(B, o_B), flush=True)
o_rT_r = orc.pprf_receiver_expand(conf, msg, choices, o_sums, o_last)
o_A, _ = orc.qc_dual_encode2(conf, o_rT_r, pts, packed=True)
print("receiver A equal:", np.array_equal(A, o_A), "total oracle %.1fs" % (time.time()-t0), flush=True)
