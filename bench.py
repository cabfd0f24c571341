#!/usr/bin/env python3
"""bench.py -- random OTs/s of the silent-OT expansion hot path on MI355X.

A "step" is one SilentMtProvider::precompute_mts equivalent (gmw/src/mul_triple/silent_ot.rs:79-114,
what flute_benchmark's `time_silent_ots_ms` covers, bench.rs:274-278) for n = 2^24 requested OTs:
one sender expansion + one receiver expansion (GGM/PPRF -> QuasiCyclic encode) + fused CR hashes and
MT-bit derivation, after the base-OT seeds exist. value = 2n * steps / time ("times two because both
parties generate this amount", bench.rs:271-272).
N > 1 (default --mode strong): that ONE step is split over the ranks. N = 2: by direction -- rank 0 runs the sender
instance, rank 1 the receiver instance, one all-gather of two bit vectors per rank joins them (RoleSplitPipeline). N = 4, 8
(or --split rows): tree sets in the GGM stage, bit rows of both roles in the QC stage, output ranges in the hash stage, two
RCCL all-to-alls between the stages and an all-gather of the MT words (csrc/shard.hip, flute_amd/dist.py). Either way the
result is bit-identical to the one-GPU result, which rank 0 checks after the timed region (--no-verify to skip). --mode weak: independent instances per rank, value
aggregated over ranks.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched through torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable copy)
QC_BYTES_PER_OT = 48.0  # 32 B read (rT) + 16 B write per requested OT
# AES stage ceiling: 160 T-table lookups (ds_read_b32) per block; a wave64 ds_read_b32 occupies the CU's LDS for
# 2 cycles (MI355X_MICROARCH.md, LDS table) -> 256 CUs x 2.4 GHz / 2 x 64 lanes / 160 lookups
AES_PEAK_PER_S = 256 * 2.4e9 / 2 * 64 / 160
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2  # wave-instructions/s: 256 CUs x 4 SIMDs, one wave64 VALU instruction per 2 cycles
GGM_KERNELS = ("k_roots", "k_ggm_level", "k_ggm_leaf", "k_sender_treegrp", "k_recv_init", "k_recv_fix", "k_recv_last")
QC_KERNELS = ("k_clear_lsb", "k_qc_load", "k_conv_fwd", "k_conv_fwd_tile", "k_fft_top", "k_fft_mid", "k_fft_low_mul",
              "k_ifft_mid", "k_ifft_top", "k_conv_inv_tile", "k_conv_inv", "k_qc_store", "k_qc_final")


def cpu_baseline(log2n_sample, T_gpu):
    """The CPU oracle (a port of the zappot/gmw path; the GF(2)[x] product runs the REFERENCE's own bitpolymul2 C when
    oracle/_ref is present) timed on this host's cores with the reference's thread policy: SilentMtProvider gives each
    direction a pool of available_parallelism() / 2 threads and runs both directions concurrently
    (gmw/src/mul_triple/silent_ot.rs:47,87-91); the GGM workers stride the tree groups (pprf.rs:128), the 128 QC rows
    and the output transposes run on the same pool, the hashes and the MT-bit loop are single-threaded as in the
    reference (silent_ot/mod.rs:275-287, silent_ot.rs:93-111)."""
    import threading
    from oracle import sot_oracle as orc
    orc.build()
    ref_mul = orc.use_reference_multiplier(True)
    n = 1 << log2n_sample
    conf = orc.configure(n)
    cores = int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1))
    cores = max(1, min(cores, os.cpu_count() or cores))
    T = max(1, cores // 2)
    rng = np.random.default_rng(123)
    x = orc.fake_base(conf, rng)
    y = orc.fake_base(conf, rng)
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    # peer's sender for direction Y (not timed): gives the TreeGrps our receiver consumes
    orc.set_num_threads(cores)
    _, sums_y, last_y = orc.pprf_sender_expand(conf, y[0], delta, seed, T)
    out = {}

    def send():
        orc.set_num_threads(T)
        out["s"] = orc.random_silent_send(conf, x[0], delta, seed, T)[0]

    def recv():
        orc.set_num_threads(T)
        r = orc.random_silent_receive(conf, y[1], y[2], sums_y, last_y)
        out["r"] = (r[0], r[1])

    t0 = time.perf_counter()
    th = [threading.Thread(target=send), threading.Thread(target=recv)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    orc.mt_bits(out["s"].reshape(-1, 2), out["r"][0], out["r"][1])
    dt = time.perf_counter() - t0
    # the same work with ONE worker per direction, directions one after the other (T = 1: what a single core does)
    orc.set_num_threads(1)
    t1 = time.perf_counter()
    s1 = orc.random_silent_send(conf, x[0], delta, seed, 1)[0]
    _, sy1, ly1 = None, sums_y, last_y
    r1 = orc.random_silent_receive(conf, y[1], y[2], sy1, ly1)
    orc.mt_bits(s1.reshape(-1, 2), r1[0], r1[1])
    dt1 = time.perf_counter() - t1
    orc.set_num_threads(cores)
    orc.use_reference_multiplier(False)
    return {
        "value": 2 * n / dt, "unit": "OTs/s", "cores": cores, "kind": "port",
        "single_thread_value": 2 * n / dt1,
        "sample": ("full precompute_mts pipeline (sender+receiver expand, QC encode, hashes, MT bits) at n=2^%d with the "
                   "reference's thread policy: both directions concurrently, T = cores/2 = %d workers each (GGM groups, "
                   "128 QC rows, transposes), hashes and MT bits single-threaded; GF(2)[x] product = %s; %.2f s wall "
                   "(one worker, directions in sequence: %.2f s)") % (
            log2n_sample, T, "reference bitpolymul_2_128 (oracle/_ref)" if ref_mul else "oracle Karatsuba+PCLMUL", dt, dt1),
    }


def aes128_workloads(fa, n, ms_per_step, lut):
    """BASELINE configs 4 and 5 in numbers: what one step's 2^k-triple device store buys, and the rates. Both paths draw
    on the same store; FLUTE's function-dependent setup needs 49 400 triples per AES-128 (200 sbox8 LUTs x 247 subset
    products), Boolean GMW 6 400 (the Bristol circuit's AND gates)."""
    mts_per_s = n / (ms_per_step * 1e-3)  # this party's triples per second at the measured step time
    out = {
        "bristol_and_gates_per_aes": fa.AES128_BRISTOL_AND_GATES,
        "bristol_source": "crates/flute_benchmark/Baseline/bristol/aes_128.bristol (36663 gates, 6400 AND)",
        "flute_sbox8_luts_per_aes": fa.AES128_SBOXES, "flute_setup_mts_per_aes": fa.AES128_LUT_SETUP_MTS,
        "gmw_aes_instances_per_store": n // fa.AES128_BRISTOL_AND_GATES,
        "flute_aes_instances_per_store": n // fa.AES128_LUT_SETUP_MTS,
        "gmw_aes_instances_per_s_mt_supply": mts_per_s / fa.AES128_BRISTOL_AND_GATES,
        "flute_aes_instances_per_s_mt_supply": mts_per_s / fa.AES128_LUT_SETUP_MTS,
    }
    if lut and lut.get("gmw_aes128") and lut["gmw_aes128"].get("value"):
        # measured, not derived: the 60 AND layers of the Bristol circuit's profile on this GPU (both parties)
        out["gmw_aes_instances_per_s_and_layers"] = lut["gmw_aes128"]["value"]
    if lut and lut.get("value"):
        # the AND layers themselves (both parties on this GPU): sbox8 instances/s measured above / 200
        out["flute_aes_instances_per_s_setup_layers"] = lut["value"] / fa.AES128_SBOXES
        out["flute_setup_batch_aes_instances"] = lut["batch"] / fa.AES128_SBOXES
        if lut.get("online") and lut["online"].get("value"):
            # the online LUT rounds of the same S-boxes (both parties, per-call transfers included) / 200
            out["flute_aes_instances_per_s_online_luts"] = lut["online"]["value"] / fa.AES128_SBOXES
    return out


def lut_online_bench(fa, cs, lam, B, reps):
    """FLUTE online LUT evaluation of the same batch on the subset products the setup layers just left in
    HBM: both parties' delta shares of the 8 outputs of every sbox8 instance, reconstructed and checked against the
    plaintext table (to_eval/sbox8.lut = the FIPS-197 S-box; tests/golden/sbox8_lut.json)."""
    path = os.path.join(ROOT, "tests", "golden", "sbox8_lut.json")
    tabs = []
    for l in json.load(open(path))["luts"]:
        v = int(l["table_hex"], 16)
        tabs.append([(v >> (255 - j)) & 1 for j in range(256)])  # Msb0 bits of the big-endian number (lut_circuit.rs:355-399)
    tmpl = fa.LutTemplate([(tuple(range(8)), t) for t in tabs])
    rng = np.random.default_rng(7)
    x = rng.integers(0, 2, (B, 8), dtype=np.uint8)
    m = x ^ lam[0].reshape(B, 8) ^ lam[1].reshape(B, 8)
    rows = tmpl.rows(m)
    o = [rng.integers(0, 2, (B, 8), dtype=np.uint8) for _ in range(2)]
    # host-buffer form once (checks the result), then the device-resident form timed: masked inputs / output-lambda shares /
    # delta shares stay in HBM, the template tables are cached in the circuit, both parties' calls are queued and waited
    # for once (sot_lut_online_eval_dev)
    d0 = fa.online_delta_shares(cs[0], tmpl, B, rows, o[0], 0)
    d1 = fa.online_delta_shares(cs[1], tmpl, B, rows, o[1], 1)
    ok = bool(np.array_equal(d0 ^ d1 ^ o[0] ^ o[1], tmpl.plain(x)))
    import torch
    t_rows = torch.from_numpy(rows.copy()).cuda()
    t_o = [torch.from_numpy(v.copy()).cuda() for v in o]
    t_d = [torch.zeros((B, tmpl.n_luts), dtype=torch.uint8, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    total = 0.0
    for rep in range(reps + 1):
        t0 = time.perf_counter()
        fa.online_delta_shares_dev(cs[0], tmpl, B, t_rows.data_ptr(), t_o[0].data_ptr(), t_d[0].data_ptr(), 0)
        fa.online_delta_shares_dev(cs[1], tmpl, B, t_rows.data_ptr(), t_o[1].data_ptr(), t_d[1].data_ptr(), 1)
        cs[0].engine.sync(), cs[1].engine.sync()
        if rep:
            total += time.perf_counter() - t0
    ok = ok and bool(np.array_equal(t_d[0].cpu().numpy(), d0) and np.array_equal(t_d[1].cpu().numpy(), d1))
    dt = total / reps
    return {"metric": "FLUTE online LUT evaluation, sbox8.lut instances/s (both parties' delta shares on one GPU)",
            "value": B / dt, "unit": "sbox8 instances/s", "lut_gates_per_s": 8 * B / dt, "batch": B, "ms_per_batch": dt * 1e3,
            "outputs_equal_plaintext_sbox": ok,
            "note": "device-resident: masked inputs, output-lambda shares and delta shares in HBM (1 byte per gate and instance "
                    "each), template tables cached in the circuit; both parties' calls queued, one wait"}


def gmw_aes128_bench(fa, eng, eng1, delta, delta_peer, n, reps):
    """BASELINE config 5: Boolean GMW on the Baseline AES-128 Bristol circuit's AND-layer profile (60 layers, 6 400 AND gates
    per instance: tests/golden/aes128_bristol_and_layers.json, computed from Baseline/bristol/aes_128.bristol by
    tools/bristol_and_layers.py; the wiring inside the profile is synthetic -- the netlist is not shipped), as many SIMD
    instances as the step's triples pay for, both parties' 60 layers in one call, triples from the same device store."""
    prof = json.load(open(os.path.join(ROOT, "tests", "golden", "aes128_bristol_and_layers.json")))["and_gates_per_layer"]
    k = n // sum(prof)
    rng = np.random.default_rng(55)
    circ = fa.gmw_and_layers([c * k for c in prof], 256 * k, rng)
    ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
    cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((eng, eng1))]
    total = 0.0
    for rep in range(reps + 1):
        eng.mt_bits(delta, download=False)
        eng1.mt_bits(delta_peer, download=False)
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        eng.sync(), eng1.sync()
        t0 = time.perf_counter()
        cs[0].run_layers_with(cs[1])
        if rep:
            total += time.perf_counter() - t0
    prod = cs[0].values() ^ cs[1].values()
    ok = bool(np.array_equal(prod[circ.n_inputs:], prod[circ.x_src] & prod[circ.y_src]))
    for c in cs:
        c.close()
    dt = total / reps
    return {"metric": "Boolean GMW AND layers of Baseline/bristol/aes_128.bristol, AES-128 instances/s (both parties on one GPU)",
            "value": k / dt, "unit": "AES-128 instances/s", "instances": k, "and_layers": len(prof), "and_gates_per_instance": sum(prof),
            "ms_per_batch": dt * 1e3, "and_gates_per_s": k * sum(prof) / dt, "products_correct": ok,
            "note": "AND-layer profile of the real circuit (180/20/40/140/100/160 per round x 10), synthetic wiring inside it"}


def lut_setup_bench(fa, eng, x, y, tg_x, delta, delta_peer, seed_peer, threads, log2b, reps):
    """FLUTE function-dependent setup (aby2_lut.rs:699-783) for `log2b` (a count) instances of to_eval/sbox8.lut (one grouped
    8-input LUT with 8 outputs = 8 LUT gates, 247 AND gates in 7 layers per instance) on the triples just expanded:
    both parties' AND layers run on this GPU (party 1 = a second engine holding the crossed OT instances), messages
    exchanged device to device. Returns the `lut_setup` object of the bench line."""
    n = eng.conf.n
    B = log2b  # sbox8 instances
    eng1 = fa.Engine(n, device=eng.device)
    try:
        eng1.sender_expand(y[0], delta_peer, seed_peer, threads)
        eng1.qc_encode(0)
        eng1.receiver_expand(x[1], x[2], *tg_x)
        eng1.qc_encode(1, eng1.points(x[2]))
        # batched circuit in circuit.interactive_iter() order: per instance (sub-circuit) its 8 LUT gates, all over the
        # same 8 parents -> 247 ANDs per instance through the subset cache
        t0 = time.perf_counter()
        circ = fa.replicate_setup_circuit(fa.build_setup_circuit([tuple(range(8))] * 8), B)
        t_build = time.perf_counter() - t0
        assert circ.n_gates == 247 * B <= n
        rng = np.random.default_rng(99)
        ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
        cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((eng, eng1))]
        total = 0.0
        for rep in range(reps + 1):
            eng.mt_bits(delta, download=False)  # refill the stores (same triples again: timing only)
            eng1.mt_bits(delta_peer, download=False)
            for c, i in zip(cs, ins):
                c.set_inputs(i)
            eng.sync(), eng1.sync()
            t0 = time.perf_counter()
            cs[0].run_layers_with(cs[1])  # all 7 layers of both parties: one fused launch per layer, one wait
            if rep:
                total += time.perf_counter() - t0
        v0, v1 = cs[0].values(), cs[1].values()
        prod = v0 ^ v1
        ok = bool(np.array_equal(prod[circ.n_inputs:], prod[circ.x_src] & prod[circ.y_src]) and
                  np.array_equal(prod[:circ.n_inputs], ins[0] ^ ins[1]))
        online = None
        try:
            online = lut_online_bench(fa, cs, ins, B, reps)
        except Exception as e:  # secondary figure
            online = {"value": None, "note": "failed: %r" % (e,)}
        for c in cs:
            c.close()
        gmw = None
        try:
            gmw = gmw_aes128_bench(fa, eng, eng1, delta, delta_peer, n, reps)
        except Exception as e:  # secondary figure
            gmw = {"value": None, "note": "failed: %r" % (e,)}
        dt = total / reps
        return {"online": online,
                "metric": "FLUTE LUT setup, sbox8.lut instances/s (both parties' AND layers on one GPU)",
                "value": B / dt, "unit": "sbox8 instances/s", "lut_gates_per_s": 8 * B / dt, "and_gates_per_s": 247 * B / dt,
                "batch": B, "ms_per_batch": dt * 1e3, "layers": int(circ.n_layers), "mts_consumed": int(circ.n_gates),
                "subset_products_correct": ok, "host_circuit_build_s": t_build,
                "gmw_aes128": gmw,
                "note": "7 layers, both parties, ONE call (sot_and_circ_run_layers_peer: a fused launch per layer, one host "
                        "synchronisation); triples come from the device MT store of the expansion above; host-side circuit "
                        "build (graph walk) not in the timed region"}
    finally:
        eng1.close()


VALU_BITOP3_PEAK = 8.0e11  # measured: tools/micro/valu_rate.hip, profiles/r1_valu_issue_rates.txt


def alsz_bench(fa, torch, device, log2n, reps):
    """ALSZ13 OT extension with device-resident contexts: receiver (2 PRG blocks per row and 128 OTs, one
    hash per OT) then sender (1 PRG block, two hashes), u rows handed over device-to-device, results left in HBM. Per
    extension: 3 AES blocks on each side; HBM bytes: receiver 16 (u) + 16 (message) + 1/8 (choice), sender 16 + 32."""
    from flute_amd import ot_ext
    count = 1 << log2n
    rng = np.random.default_rng(5)
    pairs = rng.integers(0, 2 ** 64, (128, 2, 2), dtype=np.uint64)
    bc = rng.integers(0, 256, 16, dtype=np.uint8)
    bcb = np.unpackbits(bc, bitorder="little")
    chosen = np.ascontiguousarray(pairs[np.arange(128), bcb])
    R = ot_ext.Receiver(pairs, device=device, max_count=count)
    S = ot_ext.Sender(chosen, bc.tobytes(), device=device, max_count=count)
    ch = torch.from_numpy(rng.integers(0, 256, count // 8, dtype=np.uint8)).cuda(device)
    torch.cuda.synchronize()

    def once():
        R.receive_random(None, count, to_host=False, choices_dev=ch.data_ptr())
        R.ctx.sync()
        t1 = time.perf_counter()
        S.send_random(None, count, to_host=False, u_dev=R.ctx.u_dev)
        S.ctx.sync()
        return t1
    once()
    tr = ts = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        t1 = once()
        t2 = time.perf_counter()
        tr += t1 - t0
        ts += t2 - t1
    # spot check on the last run: message of OT i on the receiver = the sender's message pair entry chosen by bit i
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    k = 4096
    rm = np.empty((k, 2), dtype=np.uint64)
    sm = np.empty((k, 2, 2), dtype=np.uint64)
    hip.hipMemcpy(rm.ctypes.data_as(C.c_void_p), C.c_void_p(R.ctx.msgs_dev), C.c_size_t(rm.nbytes), 2)
    hip.hipMemcpy(sm.ctypes.data_as(C.c_void_p), C.c_void_p(S.ctx.msgs_dev), C.c_size_t(sm.nbytes), 2)
    bits = np.unpackbits(ch[:k // 8].cpu().numpy(), bitorder="little")
    ok = bool(np.array_equal(rm, sm[np.arange(k), bits]))
    R.ctx.close()
    S.ctx.close()
    ms_r, ms_s = tr / reps * 1e3, ts / reps * 1e3
    aes_r = 3.0 * count / (ms_r * 1e-3)
    aes_s = 3.0 * count / (ms_s * 1e-3)
    return {
        "metric": "ALSZ13 OT extensions/s, receiver then sender on one GPU, device-resident (zappot ot_ext, n = 2^%d)" % log2n,
        "value": count / ((ms_r + ms_s) * 1e-3), "unit": "OTs/s", "count": count,
        "receiver_ms": ms_r, "sender_ms": ms_s, "receiver_equals_sender_choice": ok,
        "receiver": {"aes_blocks_per_s": aes_r, "frac_of_lds_lookup_peak": aes_r / AES_PEAK_PER_S,
                     "hbm_bytes_per_ot": 32.125, "hbm_gb_per_s": 32.125 * count / (ms_r * 1e-3) / 1e9},
        "sender": {"aes_blocks_per_s": aes_s, "frac_of_lds_lookup_peak": aes_s / AES_PEAK_PER_S,
                   "hbm_bytes_per_ot": 48.0, "hbm_gb_per_s": 48.0 * count / (ms_s * 1e-3) / 1e9},
        "bound": "lds (AES T-table lookups: 3 blocks per extension per side); the 128 x 128 bit transposes run in the wave's registers",
        "note": "wall time of one call per side incl. the upload of the 256 / 128 base-OT keys (expanded to round keys inside the kernel); one-shot "
                "sot_alsz_send / sot_alsz_receive add device allocation and PCIe copies of the u rows and messages",
    }


def load_pmc():
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (bench.py cannot run
    the profiler on itself; tools/profile.sh + tools/prof_summary.py produce profiles/pmc_latest.json)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        return None
    # the counters describe the kernels of the commit they were collected on: compare the source hashes stamped into the
    # file (tools/prof_summary.py) with the sources of this tree, and report nothing counter-derived when they differ
    import hashlib
    stale = []
    for fn, want in (pmc.get("source_sha256") or {"qc.hip": None}).items():
        try:
            with open(os.path.join(ROOT, "flute_amd", "csrc", fn), "rb") as fh:
                got = hashlib.sha256(fh.read()).hexdigest()
        except OSError:
            got = None
        if got != want:
            stale.append(fn)
    if stale:
        return {"stale": True, "note": "profiles/pmc_latest.json (commit %s) was collected on other versions of %s: the "
                                       "counter-derived fields are omitted" % (pmc.get("source_commit"), ", ".join(stale))}
    pmc["note"] = "%s; counters collected at commit %s (source hashes match this tree)" % (pmc.get("note"), pmc.get("source_commit"))
    return pmc


def roles_main(args, fa, fd, dist, torch, rank, local_rank, world, n):
    """N = 2 by direction (default there): rank 0 runs the party's sender instance, rank 1 its receiver instance, on
    ordinary one-GPU contexts; the only traffic is one all-gather of two bit vectors per rank (RoleSplitPipeline)."""
    eng = fa.Engine(n, device=local_rank)
    conf = eng.conf
    rng = np.random.default_rng(fd.instance_seed(0))  # the SAME instance on both ranks
    x = fa.fake_base(conf, rng)
    y = fa.fake_base(conf, rng)
    delta, seed, delta_peer, seed_peer = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    # the peer party's TreeGrps for direction Y (received over mpc_channel in the reference): only rank 1 needs them
    tg_y = eng.sender_expand(y[0], delta_peer, seed_peer, args.threads) if rank == 1 else (None, None)
    pts = eng.points(y[2])
    pipe = fd.RoleSplitPipeline(eng, rank)
    res = {}

    def step():
        pipe.direction(x[0], delta, seed, args.threads, y[1], y[2], tg_y[0], tg_y[1], pts)
        if not args.no_gather:
            res["mt"] = pipe.join()

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        fd.barrier(world)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = fd.max_over_ranks(time.perf_counter() - t0, world, device="cuda")
    prof = {}
    if not args.no_prof:
        eng.prof_reset()
        eng.prof_enable(True)
        barrier()
        for _ in range(args.steps):
            step()
        barrier()
        eng.prof_enable(False)
        prof = eng.prof_read()
    verified = None
    if not args.no_verify and not args.no_gather:
        ok = 1
        if rank == 0:  # this GPU's context is a whole one: run both directions on it and compare the joined store
            tg_full = eng.sender_expand(y[0], delta_peer, seed_peer, args.threads)
            eng.sender_expand(x[0], delta, seed, args.threads)
            eng.qc_encode(0)
            eng.receiver_expand(y[1], y[2], *tg_full)
            eng.qc_encode(1, pts)
            want = np.stack(eng.mt_bits(delta))
            ok = int(np.array_equal(res["mt"].cpu().numpy().view(np.uint64), want))
        t = torch.tensor([ok], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        verified = bool(t.item())
    if rank == 0:
        out = {
            "metric": "random OTs/sec for 2^%d silent-OT expansion (sender+receiver per party, MT bits)" % args.log2n,
            "value": 2.0 * n * args.steps / dt, "unit": "OTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 bitwise (GF(2)/AES, 128-bit blocks)", "data": "synthetic",
            "config": {"workload": "zappot silent OT, QuasiCyclic scaler 2, ONE party's n=2^%d precompute_mts (sender + "
                                   "receiver instance, MT bits) over 2 GPUs" % args.log2n,
                       "P": conf.P, "trees": conf.parts, "depth": conf.depth, "threads_T": args.threads,
                       "sharding": "by direction: rank 0 the sender instance, rank 1 the receiver instance (all trees, all 128 "
                                   "bit rows each); all-gather of 2 bit vectors per rank (%d B each way per step)%s" % (
                                       2 * ((n + 63) // 64) * 8, "" if not args.no_gather else " (skipped)")},
            "sharded_equals_one_gpu": verified,
        }
        if prof:
            per_step = {k: (ms / args.steps, cnt // args.steps) for k, (ms, cnt) in prof.items()}
            qc_ms = sum(ms for k, (ms, _) in per_step.items() if k in QC_KERNELS)
            dom_name, (dom_ms, dom_cnt) = max(per_step.items(), key=lambda kv: kv[1][0])
            achieved = QC_BYTES_PER_OT * n / (qc_ms * 1e-3) / 1e9 if qc_ms else None
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                "kernel": dom_name, "kernel_avg_ms": dom_ms / max(dom_cnt, 1), "kernel_launches_per_step": dom_cnt,
                "basis": "rank 0 (sender direction), QC-encode stage: 48 B/OT x n / summed device time of the stage's "
                         "kernels per step (%.3f ms); traffic not collected in multi-GPU runs" % qc_ms}
            out["kernels_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0])}
        print(json.dumps(out), flush=True)
    eng.close()
    dist.destroy_process_group()
    if verified is False:  # a number from a wrong result is no number
        raise SystemExit("bench.py: the joined MT words differ from the one-GPU context (sharded_equals_one_gpu = false)")


def strong_main(args, fa, fd, dist, torch, rank, local_rank, world, n):
    """N > 1, one expansion over all ranks (BASELINE config 3). Every rank holds the same inputs (one party's base OTs,
    delta, seeds: in the reference they come from the base-OT phase and the party's RNG, both outside the hot path)."""
    replicate = args.ggm == "replicate" or (args.ggm == "auto" and world == 2)
    eng = fa.ShardedEngine(n, rank, world, device=local_rank, ggm_replicated=replicate)
    conf = eng.conf
    rng = np.random.default_rng(fd.instance_seed(0))  # the SAME instance on every rank
    x = fa.fake_base(conf, rng)
    y = fa.fake_base(conf, rng)
    delta, seed, delta_peer, seed_peer = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    # the peer's TreeGrps for direction Y (received over mpc_channel in the reference), for the tree sets this rank expands
    tg_y = eng.sender_expand(y[0], delta_peer, seed_peer, args.threads)
    pts = eng.points(y[2])
    pipe = fd.ShardedPipeline(eng, ordered=not args.no_ordered, overlap=not args.no_overlap)
    xch = {"ms": 0.0, "on": False}
    base_exchange = pipe.exchange

    def timed_exchange(s, r):  # wall time of the collectives (stream-synchronous: the engine waits for them anyway)
        if not xch["on"]:
            return base_exchange(s, r)
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = base_exchange(s, r)
        torch.cuda.synchronize()
        xch["ms"] += (time.perf_counter() - t) * 1e3
        return out
    pipe.exchange = timed_exchange
    res = {}

    def step():
        pipe.expand(x[0], delta, seed, args.threads, y[1], y[2], *tg_y)
        pipe.exchange_leaves()
        pipe.encode(pts)
        res["mt"] = pipe.mt_bits(delta, gather=not args.no_gather)

    def barrier():
        pipe.sync()
        torch.cuda.synchronize()
        fd.barrier(world)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = fd.max_over_ranks(time.perf_counter() - t0, world, device="cuda")
    prof, dt_prof = {}, None
    if not args.no_prof:
        eng.prof_reset()
        eng.prof_enable(True)
        xch["on"] = True
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_prof = time.perf_counter() - t1
        eng.prof_enable(False)
        xch["on"] = False
        prof = eng.prof_read()
    verified = None
    if not args.no_verify and not args.no_gather:
        # bit-for-bit against the one-GPU context on rank 0's GPU (every rank holds the same gathered words)
        ok = 1
        if rank == 0:
            ref = fa.Engine(n, device=local_rank)
            try:
                tg_full = ref.sender_expand(y[0], delta_peer, seed_peer, args.threads)  # the peer's TreeGrps, all groups
                ref.sender_expand(x[0], delta, seed, args.threads)
                ref.qc_encode(0)
                ref.receiver_expand(y[1], y[2], *tg_full)
                ref.qc_encode(1, pts)
                want = np.stack(ref.mt_bits(delta))
            finally:
                ref.close()
            got = res["mt"].cpu().numpy().view(np.uint64)[:, :want.shape[1]]
            ok = int(np.array_equal(got, want))
        t = torch.tensor([ok], dtype=torch.int64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        verified = bool(t.item())
    if rank == 0:
        info = eng.info
        out = {
            "metric": "random OTs/sec for 2^%d silent-OT expansion (sender+receiver per party, MT bits)" % args.log2n,
            "value": 2.0 * n * args.steps / dt, "unit": "OTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_profiled": dt_prof / args.steps * 1e3 if dt_prof else None, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 bitwise (GF(2)/AES, 128-bit blocks)", "data": "synthetic",
            "config": {"workload": "zappot silent OT, QuasiCyclic scaler 2, ONE n=2^%d expansion (sender + receiver "
                                   "instance, MT bits) sharded over %d GPUs" % (args.log2n, world),
                       "P": conf.P, "trees": conf.parts, "depth": conf.depth, "threads_T": args.threads,
                       "sharding": "GGM: %s; QC: %d bit rows of both roles per rank; hash/MT: %d outputs per rank; "
                                   "all-to-all #1 %d B, #2 %d B per peer and step; MT words all-gathered%s" % (
                                       "replicated on every rank (no exchange #1)" if replicate else
                                       "tree sets %d..%d of %d on rank 0" % (info.set0, info.set0 + info.nsets_local - 1,
                                                                             conf.parts // 8),
                                       info.rows_per_rank, info.chunk, info.x1_bytes_per_peer, info.x2_bytes_per_peer,
                                       "" if not args.no_gather else " (skipped)")},
            "sharded_equals_one_gpu": verified,
            "pipeline": {"stream_ordered": pipe.ordered, "exchange1_sender_half_overlapped": pipe.overlap,
                         "host_syncs_per_step": 1 if pipe.ordered else 8},
        }
        if prof:
            per_step = {k: (ms / args.steps, cnt // args.steps) for k, (ms, cnt) in prof.items()}
            qc_ms = sum(ms for k, (ms, _) in per_step.items() if k in QC_KERNELS)
            dom_name, (dom_ms, dom_cnt) = max(per_step.items(), key=lambda kv: kv[1][0])
            # this rank's share of the algorithmic bytes: 48 B/OT x n x 2 directions / world, in ONE pass of the stage
            alg = QC_BYTES_PER_OT * n * 2.0 / world
            achieved = alg / (qc_ms * 1e-3) / 1e9 if qc_ms else None
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": None,
                "kernel": dom_name, "kernel_avg_ms": dom_ms / max(dom_cnt, 1), "kernel_launches_per_step": dom_cnt,
                "basis": "rank 0, QC-encode stage: 48 B/OT x n x 2 directions / %d ranks / summed device time of the "
                         "stage's kernels per step (%.3f ms); traffic not collected in multi-GPU runs" % (world, qc_ms)}
            ggm_ms = sum(ms for k, (ms, _) in per_step.items() if k in GGM_KERNELS)
            out["stages"] = {"ggm_expand_ms": ggm_ms, "qc_encode_ms": qc_ms,
                             "shard_pack_unpack_ms": sum(per_step.get(k, (0.0, 0))[0] for k in ("k_shard_pack", "k_shard_unpack")),
                             "hash_mt_bits_ms": per_step.get("k_mt_bits", (0.0, 0))[0],
                             "exchanges_wall_ms": xch["ms"] / args.steps}
            out["kernels_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(per_step.items(), key=lambda kv: -kv[1][0])}
        print(json.dumps(out), flush=True)
    pipe.close()
    eng.close()
    dist.destroy_process_group()
    if verified is False:  # a number from a wrong result is no number
        raise SystemExit("bench.py: the gathered MT words differ from the one-GPU context (sharded_equals_one_gpu = false)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=24, help="requested OTs per direction = 2^log2n")
    ap.add_argument("--threads", type=int, default=4,
                    help="rayon pool size the reference would use (roots depend on it, pprf.rs:115)")
    ap.add_argument("--cpu-log2n", type=int, default=24,
                    help="CPU baseline sample size (2^24 = the GPU workload itself: about 1.3 s on 16 cores with the "
                         "reference's thread policy + about 13 s for the one-worker leg)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true",
                    help="N > 1: skip the RCCL all-gather that reassembles the party's MT store on every GPU")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (rehearsals on a 1-GPU box)")
    ap.add_argument("--lut-log2b", type=int, default=16, help="FLUTE setup section (0 = skip)")
    ap.add_argument("--alsz-log2n", type=int, default=24, help="ALSZ13 OT-extension section (0 = skip)")
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--mode", choices=("strong", "weak"), default="strong",
                    help="N > 1: strong = ONE expansion sharded over the ranks (default), weak = one instance per rank")
    ap.add_argument("--ggm", choices=("auto", "shard", "replicate"), default="auto",
                    help="strong mode: tree sets sharded over the ranks + exchange #1, or every rank expands all trees "
                         "(auto: replicate at 2 ranks, where one xGMI link would carry the whole leaf exchange)")
    ap.add_argument("--split", choices=("auto", "rows", "roles"), default="auto",
                    help="strong mode: rows = bit rows of both roles sharded over the ranks (2, 4, 8 ranks), roles = one "
                         "direction per GPU (2 ranks; auto picks it there: nothing of the expansion crosses the link)")
    ap.add_argument("--sync-stages", action="store_true",
                    help="one GPU: every stage call waits for the GPU before it returns (round 2's behaviour)")
    ap.add_argument("--no-ordered", action="store_true",
                    help="strong mode: every stage call waits for the GPU (round 2's behaviour) instead of the stream-ordered pipeline")
    ap.add_argument("--no-overlap", action="store_true",
                    help="strong mode: exchange #1 as one all-to-all after both expansions instead of the sender's half "
                         "travelling while the receiver's trees expand")
    ap.add_argument("--no-verify", action="store_true",
                    help="strong mode: skip the bit-for-bit check of the gathered MT words against a one-GPU context")
    args = ap.parse_args()

    from flute_amd import dist as fd
    rank, local_rank, world = fd.env_world()
    if world != args.gpus and world > 1:
        args.gpus = world

    import torch
    if args.device is not None:
        local_rank = args.device
    torch.cuda.set_device(local_rank)
    dist = fd.init(args.backend, rank, world,
                   device_id=torch.device("cuda", local_rank) if args.backend == "nccl" else None)

    import flute_amd as fa
    n = 1 << args.log2n
    if world > 1 and args.mode == "strong":
        if args.split == "roles" or (args.split == "auto" and world == 2):
            if world != 2:
                raise SystemExit("--split roles needs exactly 2 ranks")
            roles_main(args, fa, fd, dist, torch, rank, local_rank, world, n)
        else:
            strong_main(args, fa, fd, dist, torch, rank, local_rank, world, n)
        return
    eng = fa.Engine(n, device=local_rank)
    conf = eng.conf
    rng = np.random.default_rng(fd.instance_seed(rank))
    x = fa.fake_base(conf, rng)  # direction X: this party sends
    y = fa.fake_base(conf, rng)  # direction Y: this party receives
    delta = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    delta_peer = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    seed_peer = rng.integers(0, 2 ** 64, 2, dtype=np.uint64)
    # the peer's TreeGrp messages for direction Y (the reference receives them over mpc_channel): produced
    # once, outside the timed region, by running the peer's sender expansion on this engine
    tg_y = eng.sender_expand(y[0], delta_peer, seed_peer, args.threads)
    pts = eng.points(y[2])
    tg_x = [None]

    # N > 1: every step ends with the one exchange of the path, the all-gather (RCCL over xGMI) that leaves the
    # party's whole MT store -- a | b | c of every rank's instance, 6 MiB per rank -- on every GPU
    gather = {"on": world > 1 and not args.no_gather, "buf": None, "note": None}

    def step():
        tg_x[0] = eng.sender_expand(x[0], delta, seed, args.threads)
        eng.qc_encode(0)
        eng.receiver_expand(y[1], y[2], *tg_y)
        eng.qc_encode(1, pts)
        eng.mt_bits(delta, download=False)
        if gather["on"]:
            try:
                _, gather["buf"] = fd.gather_mt_store(eng, world, gather["buf"])
            except Exception as e:  # same code on every rank: an API-level failure is symmetric, so all ranks fall back
                gather["on"], gather["note"] = False, "all-gather failed and was disabled: %r" % (e,)

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        fd.barrier(world)

    # stream-ordered context (sot_ctx_set_async): the stage calls that return nothing to the host do not wait for the GPU, so
    # the launches of a stage are queued while the previous stage still runs; the step's one host wait is the TreeGrp
    # download of the sender's expansion (the payload the peer party needs), the timed region ends with a full sync
    eng.set_async(not args.sync_stages)
    for _ in range(args.warmup):
        step()
    # timed region: EXACTLY args.steps steps, no per-kernel events (bracketing all ~110 launches of a step with HIP
    # events costs about 3 % of the step)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    dt = fd.max_over_ranks(dt, world, device="cuda")
    # second region of the same length, with HIP events around every launch on the context's stream: per-kernel
    # device time for the roofline figures (ms_per_step_profiled reports what this instrumentation costs)
    prof = {}
    dt_prof = None
    if not args.no_prof:
        eng.prof_reset()
        eng.prof_enable(True)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        dt_prof = time.perf_counter() - t1
        eng.prof_enable(False)
        prof = eng.prof_read()
    eng.set_async(False)

    # size-independent correctness property on the last step (check_correlated, silent_ot/mod.rs:744-777
    # cannot be used here because X and Y are different OT instances; the MT relation needs the peer).
    # The per-instance property that holds: receiver A == sender-of-Y's B xor c*delta_peer -- checked in tests.

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = fd.aggregate_ots_per_s(n, args.steps, dt, world)
        out = {
            "metric": "random OTs/sec for 2^%d silent-OT expansion (sender+receiver per party, MT bits)" % args.log2n,
            "value": value, "unit": "OTs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_profiled": dt_prof / args.steps * 1e3 if dt_prof else None, "higher_is_better": True,
            "scaling": "weak" if world > 1 else "strong", "vs_baseline": None,
            "dtype": "u32 bitwise (GF(2)/AES, 128-bit blocks)", "data": "synthetic",
            "config": {"workload": "zappot silent OT, QuasiCyclic scaler 2, n=2^%d random OTs per direction, "
                                   "sender+receiver expansion + MT bits on each GPU" % args.log2n,
                       "P": conf.P, "trees": conf.parts, "depth": conf.depth, "threads_T": args.threads,
                       "sharding": "single GPU" if world == 1 else ("independent instances per rank; MT store (a | b | c, 3n bits per rank) all-gathered over "
                                    "RCCL every step" if gather["on"] else "independent instances per rank (no collective)")},
        }
        if gather["note"]:
            out["config"]["gather_note"] = gather["note"]
        if prof:
            per_step = {k: (ms / args.steps, cnt // args.steps) for k, (ms, cnt) in prof.items()}
            qc_ms = sum(ms for k, (ms, _) in per_step.items() if k in QC_KERNELS)
            dom = max(per_step.items(), key=lambda kv: kv[1][0])
            dom_name, (dom_ms, dom_cnt) = dom
            # QC encode stage, per direction: algorithmic bytes = 48 B x n; two directions per step
            achieved = (QC_BYTES_PER_OT * n) / ((qc_ms / 2.0) * 1e-3) / 1e9
            out["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel": dom_name, "kernel_avg_ms": dom_ms / max(dom_cnt, 1), "kernel_launches_per_step": dom_cnt,
                # the same 48 B/OT x n of one direction against the dominant kernel's launch alone
                "dominant_kernel_achieved": (QC_BYTES_PER_OT * n) / (dom_ms / max(dom_cnt, 1) * 1e-3) / 1e9,
                "dominant_kernel_frac": (QC_BYTES_PER_OT * n) / (dom_ms / max(dom_cnt, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "basis": "QC-encode stage: 48 B/OT x n per direction / summed device time of the stage's kernels "
                         "per direction (%.3f ms); dominant kernel named" % (qc_ms / 2.0),
            }
            # every stage against the roofline that bounds it
            ggm_ms = sum(ms for k, (ms, _) in per_step.items() if k in GGM_KERNELS)
            aes_ggm = conf.parts * ((2 << conf.depth) - 2) + conf.parts * ((2 << conf.depth) - 4)  # sender + receiver
            mt_ms = per_step.get("k_mt_bits", (0.0, 0))[0]
            stages = {
                "ggm_expand": {"bound": "lds (AES T-table lookups)", "ms_per_step": ggm_ms,
                               "achieved": aes_ggm / (ggm_ms * 1e-3) if ggm_ms else None, "peak": AES_PEAK_PER_S,
                               "unit": "AES blocks/s"},
                "qc_encode": {"bound": "hbm", "ms_per_step": qc_ms, "achieved": achieved, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s (48 B/OT algorithmic)"},
                "hash_mt_bits": {"bound": "lds (AES T-table lookups)", "ms_per_step": mt_ms,
                                 "achieved": 3.0 * n / (mt_ms * 1e-3) if mt_ms else None, "peak": AES_PEAK_PER_S,
                                 "unit": "AES blocks/s"},
            }
            for st in stages.values():
                st["frac"] = st["achieved"] / st["peak"] if st["achieved"] else None
            out["stages"] = stages
            pmc = load_pmc()
            if pmc and pmc.get("stale"):
                out["roofline"]["traffic_note"] = pmc["note"]
                pmc = None
            if pmc:
                out["roofline"]["traffic"] = pmc.get("dominant_traffic_bytes_per_launch")
                out["roofline"]["traffic_note"] = pmc.get("note")
                if pmc.get("qc_stage_traffic_bytes_per_direction") and n == 1 << 24:
                    # the whole QC-encode stage of one direction, summed over its launches from the same committed passes
                    # (FETCH_SIZE corrected per access pattern, tools/prof_summary.py) against the 48 B/OT it has to move
                    st = pmc["qc_stage_traffic_bytes_per_direction"]
                    out["roofline"]["stage_traffic_bytes"] = st
                    out["roofline"]["stage_algorithmic_bytes"] = QC_BYTES_PER_OT * n
                    out["roofline"]["stage_traffic_ratio"] = st / (QC_BYTES_PER_OT * n)
                    out["roofline"]["stage_traffic_gb_per_s"] = st / ((qc_ms / 2.0) * 1e-3) / 1e9
                if pmc.get("ggm_valu_insts_per_s") and n == 1 << 24:
                    # GGM stage against the hardware's own peaks (counters of the committed pass): VALU issue = one wave
                    # instruction per 2 cycles and SIMD; LDS busy = 2 cycles of the CU's LDS pipe per wave64 ds_read_b32
                    stages["ggm_expand"]["valu_issue_frac_of_peak"] = pmc["ggm_valu_insts_per_s"] / VALU_ISSUE_PEAK
                    stages["ggm_expand"]["lds_busy_frac"] = pmc["ggm_lds_insts_per_s"] * 2.0 / (256 * 2.4e9)
                    stages["ggm_expand"]["counters_source"] = pmc.get("source")
                if pmc.get("valu_insts_per_launch") and n == 1 << 24:
                    # what actually bounds the dominant kernel: VALU issue. Wave-instructions per
                    # launch (PMC SQ_INSTS_VALU, same committed pass) / measured launch time, against the measured
                    # chip-wide ceiling for v_bitop3_b32 with three VGPR operands (profiles/r1_valu_issue_rates.txt)
                    rate = pmc["valu_insts_per_launch"] / (out["roofline"]["kernel_avg_ms"] * 1e-3)
                    out["roofline"]["dominant_kernel_valu_issue"] = {
                        "achieved": rate, "peak": VALU_BITOP3_PEAK, "unit": "wave-instructions/s",
                        "frac": rate / VALU_BITOP3_PEAK}
            out["kernels_ms_per_step"] = {k: round(v[0], 4) for k, v in sorted(per_step.items(),
                                                                                key=lambda kv: -kv[1][0])}
        # FLUTE setup section: as many whole AES-128 instances (200 sbox8 LUTs each) as the step's triples pay for
        lut_batch = fa.AES128_SBOXES * (n // fa.AES128_LUT_SETUP_MTS) if args.lut_log2b else 0
        if lut_batch and world == 1:
            try:
                out["lut_setup"] = lut_setup_bench(fa, eng, x, y, tg_x[0], delta, delta_peer, seed_peer, args.threads,
                                                   lut_batch, 5)
            except Exception as e:  # a secondary section: never takes the headline line down
                out["lut_setup"] = {"value": None, "note": "failed: %r" % (e,)}
        out["aes128"] = aes128_workloads(fa, n, ms_per_step, out.get("lut_setup"))
        if args.alsz_log2n and world == 1:
            try:
                out["alsz"] = alsz_bench(fa, torch, local_rank, args.alsz_log2n, 5)
            except Exception as e:
                out["alsz"] = {"value": None, "note": "failed: %r" % (e,)}
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_log2n, args.threads)
            except Exception as e:  # the baseline is a report, never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "OTs/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
