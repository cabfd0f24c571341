"""GPU test of the FLUTE online LUT evaluation (sot_lut_online_eval) on the subset products the setup AND
layers left in HBM: both parties' delta shares equal oracle/lut_eval_oracle.py (the reference's compute_delta_share with
the fold repaired) and reconstruct the plaintext LUT outputs -- for to_eval/sbox8.lut that is the FIPS-197 S-box."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def fa():
    import flute_amd
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests need the HIP path")
    return flute_amd


def _two_providers(fa, n, T, seed):
    from tests.test_gpu_lut_setup import _two_providers as tp
    return tp(fa, n, T, seed)


def _run_setup(fa, e0, e1, circ, lam0, lam1):
    c0, c1 = fa.AndCircuit(e0, circ, 0), fa.AndCircuit(e1, circ, 1)
    c0.set_inputs(lam0)
    c1.set_inputs(lam1)
    for layer in range(circ.n_layers):
        c0.layer_msg(layer, download=False)
        c1.layer_msg(layer, download=False)
        c0.layer_eval_from(layer, c1)
        c1.layer_eval_from(layer, c0)
    return c0, c1


def _sbox8_template(fa):
    from oracle import lut_eval_oracle as le
    d = json.load(open(os.path.join(HERE, "golden", "sbox8_lut.json")))
    return fa.LutTemplate([(tuple(range(8)), le.table_bits(l["table_hex"], 8)) for l in d["luts"]])


def _check(fa, tmpl, batch, n, seed, oracle_instances):
    from oracle import lut_eval_oracle as le
    e0, e1, _ = _two_providers(fa, n, 2, seed)
    try:
        circ = tmpl.setup_circuit(batch)
        ni = tmpl.setup.n_inputs
        rng = np.random.default_rng(seed + 1)
        x = rng.integers(0, 2, (batch, ni), dtype=np.uint8)  # plaintext wire values
        lam0 = rng.integers(0, 2, (batch, ni), dtype=np.uint8)
        lam1 = rng.integers(0, 2, (batch, ni), dtype=np.uint8)
        m = x ^ lam0 ^ lam1  # public masked values (Share::public of the input wires)
        c0, c1 = _run_setup(fa, e0, e1, circ, lam0.reshape(-1), lam1.reshape(-1))
        rows = tmpl.rows(m)
        o0 = rng.integers(0, 2, (batch, tmpl.n_luts), dtype=np.uint8)
        o1 = rng.integers(0, 2, (batch, tmpl.n_luts), dtype=np.uint8)
        d0 = fa.online_delta_shares(c0, tmpl, batch, rows, o0, 0)
        d1 = fa.online_delta_shares(c1, tmpl, batch, rows, o1, 1)
        # the protocol: Delta_out ^ lambda_out = the LUT's value on the plaintext inputs, every instance and gate
        assert np.array_equal(d0 ^ d1 ^ o0 ^ o1, tmpl.plain(x))
        # the device-resident form (sot_lut_online_eval_dev: per-item arrays in HBM, tables cached, no wait): same bits
        import torch
        t_rows, t_o0 = torch.from_numpy(rows.copy()).cuda(), torch.from_numpy(o0.copy()).cuda()
        t_out = torch.full((batch, tmpl.n_luts), 7, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        fa.online_delta_shares_dev(c0, tmpl, batch, t_rows.data_ptr(), t_o0.data_ptr(), t_out.data_ptr(), 0)
        fa.online_delta_shares_dev(c0, tmpl, batch, t_rows.data_ptr(), t_o0.data_ptr(), t_out.data_ptr(), 0)  # cached tables
        e0.sync()
        assert np.array_equal(t_out.cpu().numpy(), d0)
        # each party's share against the restated compute_delta_share, fed with the SAME setup shares
        # setup() sorts a gate's input shares before taking the powerset (aby2_lut.rs:735), compute_delta_share walks the
        # powerset of the inputs in GATE order: the two orders agree when a gate lists its wires in ascending setup-input
        # order (every gate of sbox8.lut). For a gate that does not, the shares are handed to the oracle re-indexed by
        # subset, which is the pairing the protocol means (and the one the kernel's lsrc table encodes).
        s0, s1 = c0.lut_eval_shares(), c1.lut_eval_shares()
        for inst in oracle_instances:
            for g, par in enumerate(tmpl.parents):
                cols = [tmpl.input_of[w] for w in par]
                li = inst * tmpl.n_luts + g
                sorted_subsets = [tuple(q) for q in le.powerset(sorted(cols))[len(cols) + 1:]]
                gate_subsets = [tuple(sorted(cols[i] for i in q)) for q in le.powerset(range(len(cols)))[len(cols) + 1:]]
                perm = [sorted_subsets.index(q) for q in gate_subsets]
                w0 = le.compute_delta_share(0, tmpl.tables[g], [int(m[inst, c]) for c in cols],
                                            [int(lam0[inst, c]) for c in cols], [int(s0[li][k]) for k in perm], int(o0[inst, g]))
                w1 = le.compute_delta_share(1, tmpl.tables[g], [int(m[inst, c]) for c in cols],
                                            [int(lam1[inst, c]) for c in cols], [int(s1[li][k]) for k in perm], int(o1[inst, g]))
                assert (int(d0[inst, g]), int(d1[inst, g])) == (w0, w1), (inst, g)
        return e0, c0, tmpl, rows, o0
    finally:
        e0.close()
        e1.close()


def test_sbox8_online_evaluation(fa):
    """300 SIMD instances of to_eval/sbox8.lut: 8 LUT gates of 8 inputs over one input set (74 100 setup triples)"""
    _check(fa, _sbox8_template(fa), 300, 100000, 5, oracle_instances=(0, 299))


def test_online_evaluation_one_thread_per_item_kernel(fa, monkeypatch):
    """the fallback kernel k_lut_online (templates too large for the cooperative kernel's LDS), selected here by the
    SOT_NO_LUT_COOP switch that is read at sot_ctx_create: same checks"""
    monkeypatch.setenv("SOT_NO_LUT_COOP", "1")
    _check(fa, _sbox8_template(fa), 70, 30000, 6, oracle_instances=(0, 69))


def test_mixed_template_online_evaluation(fa):
    """gates of 1 .. 6 inputs, overlapping and permuted input sets (the subset cache and the sorted-input pairing)"""
    rng = np.random.default_rng(3)
    luts = []
    for par in [(0, 1), (2, 1, 0), (3,), (1, 4, 2, 0), (5, 4, 3, 2, 1), (0, 5, 1, 4, 2, 3), (4, 5)]:
        luts.append((par, [int(b) for b in rng.integers(0, 2, 1 << len(par))]))
    luts[0] = ((0, 1), [0, 0, 0, 1])  # LutGate::and (aby2_lut.rs:372-377)
    _check(fa, fa.LutTemplate(luts), 129, 20000, 9, oracle_instances=(0, 77, 128))


def test_online_evaluation_refuses_bad_calls(fa):
    from flute_amd.lib import SotError
    tmpl = _sbox8_template(fa)
    e0, e1, _ = _two_providers(fa, 3000, 2, 2)
    try:
        circ = tmpl.setup_circuit(2)
        c0 = fa.AndCircuit(e0, circ, 0)
        c0.set_inputs(np.zeros(circ.n_inputs, dtype=np.uint8))
        z = np.zeros((2, 8), dtype=np.uint8)
        with pytest.raises(SotError):
            fa.online_delta_shares(c0, tmpl, 2, z, z, 0)  # setup layers not evaluated
        with pytest.raises(SotError):
            fa.online_delta_shares(c0, tmpl, 3, np.zeros((3, 8), np.uint8), np.zeros((3, 8), np.uint8), 0)  # wrong batch
    finally:
        e0.close()
        e1.close()


def test_aes128_flute_end_to_end_fips197(fa):
    """BASELINE config 4 end to end on the GPU: silent-OT triples -> function-dependent setup of all 200 S-boxes of two
    AES-128 instances (98 800 triples) -> ten online LUT rounds (Delta shares exchanged per round) -> the FIPS-197
    appendix C.1 ciphertext for instance 0 and a random key / plaintext pair checked against a plain AES for instance 1.
    (Also a CPU cross-check of the composition itself: the same walk with a table S-box gives the same bytes.)"""
    from flute_amd import aes_lut as al
    from oracle import lut_eval_oracle as le
    from tests.test_lut_eval import _aes_sbox
    d = json.load(open(os.path.join(HERE, "golden", "sbox8_lut.json")))
    tabs = [le.table_bits(l["table_hex"], 8) for l in d["luts"]]
    batch = 2
    aes = al.Aes128Flute(tabs, batch)
    assert aes.setup_circuit.n_gates == batch * fa.AES128_LUT_SETUP_MTS
    rng = np.random.default_rng(2024)
    key = np.stack([np.arange(16, dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)])
    pt = np.stack([np.array([0x11 * i for i in range(16)], dtype=np.uint8), rng.integers(0, 256, 16, dtype=np.uint8)])
    S = np.array(_aes_sbox(), dtype=np.uint8)
    want = al.bits_to_bytes(al.aes128_walk(al.bytes_to_bits(pt), al.bytes_to_bits(key),
                                           lambda r, x: al.bytes_to_bits(S[al.bits_to_bytes(x)]), True))
    assert want[0].tobytes().hex() == "69c4e0d86a7b0430d8cdb78070b4c55a"  # FIPS-197 C.1
    e0, e1, _ = _two_providers(fa, 100000, 2, 77)
    try:
        lam_in = aes.offline(rng)
        c0, c1 = _run_setup(fa, e0, e1, aes.setup_circuit, lam_in[0], lam_in[1])
        assert e0.lib.sot_mt_remaining(e0.h) == 100000 - batch * fa.AES128_LUT_SETUP_MTS
        got = aes.online([c0, c1], pt, key)
        assert aes.rounds_run == 10
        assert np.array_equal(got, want), (got[0].tobytes().hex(), want[0].tobytes().hex())
    finally:
        e0.close()
        e1.close()
