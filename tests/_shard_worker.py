"""One rank of a multi-process run of the sharded / role-split pipelines (started by tests/test_gpu_zy_rccl.py as a fresh
child process, one per rank): ONE expansion over `world` ranks through torch.distributed -- backend "nccl" = RCCL with
one GPU per rank, or "gloo" with every rank on GPU 0 (the transport then stages through the host; everything else is the
same code) -- and rank 0 checks the gathered MT words, B, A and hashed messages bit for bit against a one-GPU context.
usage: _shard_worker.py <mode shard|roles> <backend> <rank> <world> <port> <n> <replicate 0|1> [ordered 0|1|2]; exit code
0 = equal. ordered: 0 = every stage call waits for the GPU, 1 = stream-ordered pipeline (sot_ctx_set_stream / _set_async),
2 = stream-ordered + the sender's half of exchange #1 overlapped with the receiver's expansion; two steps are run so that
the second one starts while nothing has been waited for."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    mode, backend = sys.argv[1], sys.argv[2]
    rank, world, port, n, replicate = (int(a) for a in sys.argv[3:8])
    ordered = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    import flute_amd as fa
    from flute_amd import dist as fd
    dist = fd.init(backend, rank, world, device_id=torch.device("cuda", dev) if backend == "nccl" else None)
    T = 4
    rng = np.random.default_rng(n % 1013)  # the SAME instance on every rank
    conf = fa.QuasiCyclicConf.configure(n)
    x, y = fa.fake_base(conf, rng), fa.fake_base(conf, rng)
    delta, seed, dp, sp = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    want = None
    if rank == 0:  # the one-GPU result (and the peer party's TreeGrps for direction Y)
        ref = fa.Engine(n, device=dev)
        tg_y = ref.sender_expand(y[0], dp, sp, T)
        ref.sender_expand(x[0], delta, seed, T)
        ref.qc_encode(0)
        ref.receiver_expand(y[1], y[2], *tg_y)
        ref.qc_encode(1, ref.points(y[2]))
        want = {"B": ref.download_encoded(0), "A": ref.download_encoded(1), "s": ref.sender_hash(delta),
                "r": ref.receiver_hash()[0], "mt": np.stack(ref.mt_bits(delta))}
        ref.close()
        tg = [torch.from_numpy(t.view(np.int64).copy()) for t in tg_y]
    else:
        s_shape, l_shape = conf.treegrp_shapes()
        tg = [torch.empty(s_shape, dtype=torch.int64), torch.empty(l_shape, dtype=torch.int64)]
    # every rank needs the peer's TreeGrps (in the reference they arrive over mpc_channel)
    for t in tg:
        if backend == "nccl":
            g = t.cuda()
            dist.broadcast(g, 0)
            t.copy_(g.cpu())
        else:
            dist.broadcast(t, 0)
    tg_y = [t.numpy().view(np.uint64) for t in tg]
    ok = True
    if mode == "roles":
        eng = fa.Engine(n, device=dev)
        pipe = fd.RoleSplitPipeline(eng, rank)
        pipe.direction(x[0], delta, seed, T, y[1], y[2], tg_y[0], tg_y[1], eng.points(y[2]))
        got = pipe.join().cpu().numpy().view(np.uint64)
        if rank == 0:
            ok = np.array_equal(got, want["mt"])
    else:
        eng = fa.ShardedEngine(n, rank, world, device=dev, ggm_replicated=bool(replicate))
        pipe = fd.ShardedPipeline(eng, ordered=ordered >= 1, overlap=ordered >= 2)
        pts = eng.points(y[2])
        for _ in range(2 if ordered else 1):
            pipe.expand(x[0], delta, seed, T, y[1], y[2], *tg_y)
            pipe.exchange_leaves()
            pipe.encode(pts)
            mt_t = pipe.mt_bits(delta, gather=True)
        mt = mt_t.cpu().numpy().view(np.uint64)
        outs = {}
        for key, what, dl in (("B", eng.OUT_B, None), ("A", eng.OUT_A, None), ("s", eng.OUT_SENDER_MSGS, delta),
                              ("r", eng.OUT_RECEIVER_MSGS, None)):
            outs[key] = pipe.gather_blocks(what, delta=dl)[0].cpu().numpy().view(np.uint64)
        pipe.sync()
        pipe.close()
        if rank == 0:
            ok = np.array_equal(mt[:, :want["mt"].shape[1]], want["mt"])
            for key in outs:
                ok = ok and np.array_equal(outs[key], want[key])
                if not np.array_equal(outs[key], want[key]):
                    print("rank 0: gathered %s differs from the one-GPU context" % key, flush=True)
    eng.close()
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) == 1 else 1)


if __name__ == "__main__":
    main()
