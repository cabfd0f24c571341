import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
import sys, time
import numpy as np
import flute_amd as fa
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log2n
T = 4
t0 = time.time()
eng = fa.Engine(n)
print("ctx create %.3fs, device bytes %.2f GiB" % (time.time() - t0, eng.device_bytes() / 2**30), flush=True)
rng = np.random.default_rng(1)
msg2, msg, choices = fa.fake_base(eng.conf, rng)
delta = rng.integers(0, 2**64, 2, dtype=np.uint64)
seed = rng.integers(0, 2**64, 2, dtype=np.uint64)
pts = eng.points(choices)
for it in range(3):
    t0 = time.time(); tg = eng.sender_expand(msg2, delta, seed, T); t1 = time.time()
    eng.qc_encode(0); t2 = time.time()
    eng.receiver_expand(msg, choices, *tg); t3 = time.time()
    eng.qc_encode(1, pts); t4 = time.time()
    eng.mt_bits(delta, download=False); t5 = time.time()
    print("it%d: s_expand %.2f ms, s_encode %.2f ms, r_expand %.2f ms, r_encode %.2f ms, mt %.2f ms, total %.2f ms -> %.3e OT/s" % (
        it, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t5-t0)*1e3, 2*n/(t5-t0)), flush=True)
# correctness property at scale: check_correlated on packed outputs
B = eng.download_encoded(0); A = eng.download_encoded(1)
c = (A[:,0] & np.uint64(1)).astype(bool)
d = np.array(delta); 
x = A ^ B
x[:,0] &= np.uint64(0xFFFFFFFFFFFFFFFE)
dm = d.copy(); dm[0] &= np.uint64(0xFFFFFFFFFFFFFFFE)
ok0 = (~c) & (x[:,0]==0) & (x[:,1]==0)
ok1 = c & (x[:,0]==dm[0]) & (x[:,1]==dm[1])
print("check_correlated:", bool((ok0|ok1).all()), "ones:", int(c.sum()), flush=True)
