"""The multi-PROCESS form of the sharded expansion, one process per rank over torch.distributed, as bench.py --gpus N
runs it (flute_amd/dist.py: all_to_all_bytes, gather_sharded_words, gather_sharded_blocks, RoleSplitPipeline._all_gather).
* backend nccl (= RCCL over xGMI), one GPU per rank, world = min(8, visible GPUs): needs >= 2 GPUs, a clean skip on one --
  so that RCCL's first execution is a test, not the benchmark;
* backend gloo with every rank on GPU 0: the same pipeline code and the same checks on a one-GPU box (only the transport
  differs: device tensors are staged through the host).
Rank 0 compares the gathered MT words and the gathered B / A / hashed messages with a one-GPU context, bit for bit.
Every rank is a FRESH child process (tests/_shard_worker.py) that sets its device before anything touches the GPU."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, backend, world, n, replicate=0, ordered=0, timeout=240):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_shard_worker.py"), mode, backend, str(r),
                               str(world), str(port), str(n), str(replicate), str(ordered)], env=env, cwd=ROOT)
             for r in range(world)]
    rcs = []
    try:
        for p in procs:
            rcs.append(p.wait(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world, "rank exit codes %r" % (rcs,)


def _gpus():
    import torch
    return torch.cuda.device_count()  # counting devices does not initialise the GPU


@pytest.mark.parametrize("world,n,replicate,ordered", [(2, 1 << 20, 0, 0), (4, 100003, 0, 1), (2, 40000, 1, 1),
                                                       (4, 1 << 20, 0, 2), (8, 3 * 2 ** 20 + 12345, 0, 2)])
def test_sharded_step_multiprocess_gloo_one_gpu(world, n, replicate, ordered):
    """ordered 1 / 2: the stream-ordered pipeline bench.py uses (no host waits between the stages; 2: the sender's half of
    exchange #1 on a second stream while the receiver's trees expand)"""
    _run("shard", "gloo", world, n, replicate, ordered)


def test_role_split_multiprocess_gloo_one_gpu():
    _run("roles", "gloo", 2, 1 << 20)


def test_sharded_step_over_rccl():
    g = _gpus()
    if g < 2:
        pytest.skip("RCCL needs one GPU per rank: %d GPU visible (the gloo variants above run the same pipeline code)" % g)
    world = 8 if g >= 8 else 4 if g >= 4 else 2
    _run("shard", "nccl", world, 1 << 20, ordered=2)  # what bench.py --gpus N runs
    _run("shard", "nccl", world, 1 << 20, ordered=0)
    _run("shard", "nccl", 2, 1 << 20, replicate=1, ordered=1)


def test_role_split_over_rccl():
    g = _gpus()
    if g < 2:
        pytest.skip("RCCL needs one GPU per rank: %d GPU visible" % g)
    _run("roles", "nccl", 2, 1 << 20)
