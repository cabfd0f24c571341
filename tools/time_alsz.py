"""wall time of the device-resident ALSZ calls (what bench.py's alsz section reports); run under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations:  python tools/time_alsz.py [log2n] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from flute_amd import ot_ext

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
count = 1 << log2n
rng = np.random.default_rng(5)
pairs = rng.integers(0, 2 ** 64, (128, 2, 2), dtype=np.uint64)
bc = rng.integers(0, 256, 16, dtype=np.uint8)
chosen = np.ascontiguousarray(pairs[np.arange(128), np.unpackbits(bc, bitorder="little")])
R = ot_ext.Receiver(pairs, device=0, max_count=count)
S = ot_ext.Sender(chosen, bc.tobytes(), device=0, max_count=count)
ch = torch.from_numpy(rng.integers(0, 256, count // 8, dtype=np.uint8)).cuda(0)
torch.cuda.synchronize()
tr = ts = 0.0
for i in range(reps + 1):
    t0 = time.perf_counter()
    R.receive_random(None, count, to_host=False, choices_dev=ch.data_ptr())
    R.ctx.sync()
    t1 = time.perf_counter()
    S.send_random(None, count, to_host=False, u_dev=R.ctx.u_dev)
    S.ctx.sync()
    t2 = time.perf_counter()
    if i:
        tr += t1 - t0
        ts += t2 - t1
print("alsz n=2^%d wall: receiver %.3f ms, sender %.3f ms" % (log2n, tr / reps * 1e3, ts / reps * 1e3))
