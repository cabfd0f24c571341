#!/usr/bin/env python3
"""Per-rank COMPUTE time of the sharded expansion, measured on one GPU with nothing else running: rank 0 of a world of
2 / 4 / 8 executes its share of a precompute_mts step (GGM on its tree sets or on all of them, pack / unpack of the
exchange buffers, QC encode of its bit rows, hashes + MT bits of its outputs) with the all-to-alls replaced by no-ops
(the receive buffers hold arbitrary bytes: timing only, results are not meaningful). Together with the exchange sizes
printed beside it this gives the projected strong-scaling curve of DESIGN.md section 8 without an 8-GPU node.

  python tools/shard_rank_time.py [log2n]      -> one JSON line"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main():
    import torch
    import flute_amd as fa
    from flute_amd import dist as fd
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    n = 1 << log2n
    out = {"n": n, "ranks": []}
    for world, repl in ((2, True), (2, False), (4, True), (4, False), (8, True), (8, False)):
        eng = fa.ShardedEngine(n, 0, world, device=0, ggm_replicated=repl)
        conf = eng.conf
        rng = np.random.default_rng(1)
        x = fa.fake_base(conf, rng)
        y = fa.fake_base(conf, rng)
        delta, seed, delta_peer, seed_peer = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
        tg_y = eng.sender_expand(y[0], delta_peer, seed_peer, 4)
        pts = eng.points(y[2])
        pipe = fd.ShardedPipeline(eng, exchange=lambda s, r: None, ordered=True, overlap=True)  # as bench.py runs it

        def step():
            pipe.expand(x[0], delta, seed, 4, y[1], y[2], *tg_y)
            pipe.exchange_leaves()
            pipe.encode(pts)
            pipe.mt_bits(delta, gather=False)

        step()
        pipe.sync()
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        pipe.sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        eng.prof_reset()
        eng.prof_enable(True)
        step()
        pipe.sync()
        eng.prof_enable(False)
        prof = {k: round(v[0], 3) for k, v in sorted(eng.prof_read().items(), key=lambda kv: -kv[1][0]) if v[0] > 0.05}
        out["ranks"].append({"world": world, "ggm": "replicated" if repl else "sharded", "compute_ms_per_step": round(ms, 3),
                             "rows_per_rank_and_role": 128 // world,
                             "exchange1_bytes_to_each_peer": eng.x1_bytes, "exchange2_bytes_to_each_peer": eng.x2_bytes,
                             "kernels_ms": prof})
        pipe.close()
        eng.close()
        del pipe
        torch.cuda.empty_cache()
    # two GPUs by direction (RoleSplitPipeline): each rank's whole step on an ordinary context
    eng = fa.Engine(n, device=0)
    conf = eng.conf
    rng = np.random.default_rng(1)
    x = fa.fake_base(conf, rng)
    y = fa.fake_base(conf, rng)
    delta, seed, delta_peer, seed_peer = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    tg_y = eng.sender_expand(y[0], delta_peer, seed_peer, 4)
    pts = eng.points(y[2])
    out["roles"] = []
    for rank in (0, 1):
        pipe = fd.RoleSplitPipeline(eng, rank, gather=lambda local: None)
        pipe.direction(x[0], delta, seed, 4, y[1], y[2], tg_y[0], tg_y[1], pts)
        eng.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            pipe.direction(x[0], delta, seed, 4, y[1], y[2], tg_y[0], tg_y[1], pts)
        eng.sync()
        out["roles"].append({"rank": rank, "direction": "sender" if rank == 0 else "receiver",
                             "compute_ms_per_step": round((time.perf_counter() - t0) / 5 * 1e3, 3),
                             "all_gather_bytes_per_rank": int(pipe.buf.numel() * 8)})
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
