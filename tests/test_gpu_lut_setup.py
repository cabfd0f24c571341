"""GPU parity of the FLUTE setup AND layers (sot_and_circ_*) against oracle/lut_setup_oracle.py: messages
and every gate value bit for bit, with the multiplication triples of two crossed silent-OT instances."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fa():
    import flute_amd
    if flute_amd.device_count() < 1:
        pytest.fail("no gfx950 device: GPU tests need the HIP path")
    return flute_amd


def _two_providers(fa, n, T, seed):
    e0, e1 = fa.Engine(n), fa.Engine(n)
    rng = np.random.default_rng(seed)
    x = fa.fake_base(e0.conf, rng)
    y = fa.fake_base(e0.conf, rng)
    d0, d1, s0, s1 = (rng.integers(0, 2 ** 64, 2, dtype=np.uint64) for _ in range(4))
    mts = []
    tg0 = e0.sender_expand(x[0], d0, s0, T)
    e0.qc_encode(0)
    tg1 = e1.sender_expand(y[0], d1, s1, T)
    e1.qc_encode(0)
    e0.receiver_expand(y[1], y[2], *tg1)
    e0.qc_encode(1, e0.points(y[2]))
    e1.receiver_expand(x[1], x[2], *tg0)
    e1.qc_encode(1, e1.points(x[2]))
    for e, d in ((e0, d0), (e1, d1)):
        a, b, c = e.mt_bits(d)
        mts.append(tuple(fa.MulTriples._bits(w, n) for w in (a, b, c)))
    return e0, e1, mts


CIRCUITS = {
    "sbox8_x8": [((i, j) for j in range(8)) for i in range(8) for _ in range(8)],  # 8 instances x 8 LUT gates
    "overlap": [(0, 1, 2), (2, 1, 3), (0, 1, 2, 3), (5, 4)],
    "ragged": [(3, 9), (1, 2, 3, 4, 5), (7,), (9, 3, 1)],  # a 1-input LUT contributes nothing
}


@pytest.mark.parametrize("name", list(CIRCUITS))
def test_and_layers_match_oracle(fa, name):
    from oracle import lut_setup_oracle as lo
    luts = [tuple(l) for l in CIRCUITS[name]]
    n = 3000
    e0, e1, mts = _two_providers(fa, n, 2, 41)
    try:
        circ = fa.build_setup_circuit(luts)
        rng = np.random.default_rng(8)
        in0 = rng.integers(0, 2, circ.n_inputs, dtype=np.uint8)
        in1 = rng.integers(0, 2, circ.n_inputs, dtype=np.uint8)
        g, _, out_ids, p0, p1, tr = (lambda r: (r[0], None, r[1], r[2], r[3], r[4]))(
            lo.run_two_party(luts, in0, in1, mts[0], mts[1]))
        c0, c1 = fa.AndCircuit(e0, circ, 0), fa.AndCircuit(e1, circ, 1)
        c0.set_inputs(in0)
        c1.set_inputs(in1)
        assert len(tr) == circ.n_layers
        used = 0
        for layer in range(circ.n_layers):
            m0, m1 = c0.layer_msg(layer), c1.layer_msg(layer)
            (od0, oe0), (od1, oe1), _ = tr[layer]
            assert np.array_equal(m0[0], od0) and np.array_equal(m0[1], oe0), "party 0 message differs"
            assert np.array_equal(m1[0], od1) and np.array_equal(m1[1], oe1), "party 1 message differs"
            c0.layer_eval(layer, *m1)
            c1.layer_eval(layer, *m0)
            used += int(circ.layer_counts[layer])
            assert e0.lib.sot_mt_remaining(e0.h) == n - used
        and_ids = [i for i, k in enumerate(g.kind) if k == "and"]
        in_ids = [i for i, k in enumerate(g.kind) if k == "in"]
        for c, p in ((c0, p0), (c1, p1)):
            v = c.values()
            assert np.array_equal(v[:circ.n_inputs], p.out[in_ids])
            assert np.array_equal(v[circ.n_inputs:], p.out[and_ids]), "gate values differ from the oracle"
        # LutEvalShares in powerset order + the protocol property
        lam = in0 ^ in1
        s0, s1 = c0.lut_eval_shares(), c1.lut_eval_shares()
        for parents, a, b, outs in zip(luts, s0, s1, circ.lut_outputs):
            assert len(a) == 2 ** len(parents) - len(parents) - 1
        v0, v1 = c0.values(), c1.values()
        prod = v0 ^ v1
        for j in range(circ.n_gates):
            x, y = int(circ.x_src[j]), int(circ.y_src[j])
            assert prod[circ.n_inputs + j] == (prod[x] & prod[y])
        assert np.array_equal(prod[:circ.n_inputs], lam)
        # an out-of-order call and running out of triples are errors, not crashes
        from flute_amd.lib import SotError
        with pytest.raises(SotError):
            c0.layer_eval(0, np.zeros(1, np.uint8), np.zeros(1, np.uint8))
        c0.close()
        c1.close()
    finally:
        e0.close()
        e1.close()


def test_device_to_device_exchange_and_exhaustion(fa):
    """layer_eval_peer (both parties on one GPU) gives the same values as the host exchange; a layer that needs more
    triples than the store holds is refused."""
    from flute_amd.lib import SotError
    luts = [tuple((i, j) for j in range(6)) for i in range(40)]  # 40 x 57 = 2280 ANDs
    n = 3000
    e0, e1, _ = _two_providers(fa, n, 1, 5)
    try:
        circ = fa.build_setup_circuit(luts)
        rng = np.random.default_rng(1)
        ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
        cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((e0, e1))]
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        for layer in range(circ.n_layers):
            cs[0].layer_msg(layer, download=False)
            cs[1].layer_msg(layer, download=False)
            cs[0].layer_eval_from(layer, cs[1])
            cs[1].layer_eval_from(layer, cs[0])
        prod = cs[0].values() ^ cs[1].values()
        for j in range(circ.n_gates):
            assert prod[circ.n_inputs + j] == (prod[int(circ.x_src[j])] & prod[int(circ.y_src[j])])
        assert e0.lib.sot_mt_remaining(e0.h) == n - circ.n_gates
        # 720 triples left: a second run of layer 0 (600 gates) fits, layer 1 (800) does not
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        cs[0].layer_msg(0, download=False)
        cs[1].layer_msg(0, download=False)
        cs[0].layer_eval_from(0, cs[1])
        cs[1].layer_eval_from(0, cs[0])
        rem = e0.lib.sot_mt_remaining(e0.h)
        with pytest.raises(SotError, match="triples"):
            cs[0].layer_msg(1, download=False)
        assert e0.lib.sot_mt_remaining(e0.h) == rem  # a refused layer consumes nothing
    finally:
        e0.close()
        e1.close()


def test_layer_calls_out_of_order_are_refused(fa):
    """One message buffer per circuit: a second layer_msg while one is pending, an evaluation of another layer than the
    pending one, a layer whose predecessor is not evaluated, and a repeated layer are ESTATE errors that consume no
    triples (sot_and_circ_layer_msg / _eval); the regular order still gives the subset products afterwards."""
    from flute_amd.lib import SotError
    luts = [(0, 1, 2, 3), (2, 3, 4)]
    n = 1000
    e0, e1, _ = _two_providers(fa, n, 1, 9)
    try:
        circ = fa.build_setup_circuit(luts)
        assert circ.n_layers >= 3
        rng = np.random.default_rng(2)
        ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
        cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((e0, e1))]
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        with pytest.raises(SotError):
            cs[0].layer_msg(1)  # layer 0 not evaluated yet
        m0 = cs[0].layer_msg(0)
        rem = e0.lib.sot_mt_remaining(e0.h)
        with pytest.raises(SotError):
            cs[0].layer_msg(1)  # message of layer 0 still pending
        with pytest.raises(SotError):
            cs[0].layer_msg(0)  # twice: would consume the layer's triples again
        with pytest.raises(SotError):
            cs[0].layer_eval(1, *m0)  # evaluation of a layer without a pending message
        with pytest.raises(SotError):
            cs[0].layer_eval_from(0, cs[1])  # the peer has no message for layer 0 yet
        assert e0.lib.sot_mt_remaining(e0.h) == rem
        m1 = cs[1].layer_msg(0)
        cs[0].layer_eval(0, *m1)
        cs[1].layer_eval(0, *m0)
        with pytest.raises(SotError):
            cs[0].layer_eval(0, *m1)  # already evaluated
        with pytest.raises(SotError):
            cs[0].layer_msg(0)  # already evaluated
        for layer in range(1, circ.n_layers):
            cs[0].layer_msg(layer, download=False)
            cs[1].layer_msg(layer, download=False)
            cs[0].layer_eval_from(layer, cs[1])
            cs[1].layer_eval_from(layer, cs[0])
        prod = cs[0].values() ^ cs[1].values()
        for j in range(circ.n_gates):
            assert prod[circ.n_inputs + j] == (prod[int(circ.x_src[j])] & prod[int(circ.y_src[j])])
    finally:
        e0.close()
        e1.close()


def test_aes128_lut_setup_config4(fa):
    """BASELINE config 4: FLUTE LUT-triple generation for a full AES-128 LUT circuit (200 sbox8.lut instances per AES,
    flute_amd.lut_setup.aes128_*), two-party, on the triples of two crossed silent-OT instances: every message and every
    subset product of one AES instance against the oracle; the store shrinks by 49 400 triples per instance."""
    from oracle import lut_setup_oracle as lo
    n = 1 << 16
    e0, e1, mts = _two_providers(fa, n, 4, 44)
    try:
        batch = 1
        luts = fa.aes128_lut_gate_parents(batch)
        circ = fa.aes128_setup_circuit(batch)
        assert circ.n_gates == fa.AES128_LUT_SETUP_MTS == 49400 and circ.n_layers == 7
        rng = np.random.default_rng(4)
        ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
        g, out_ids, p0, p1, tr = lo.run_two_party(luts, ins[0], ins[1], mts[0], mts[1])
        cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((e0, e1))]
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        for layer in range(circ.n_layers):
            m0, m1 = cs[0].layer_msg(layer), cs[1].layer_msg(layer)
            (od0, oe0), (od1, oe1), _ = tr[layer]
            assert np.array_equal(m0[0], od0) and np.array_equal(m0[1], oe0)
            assert np.array_equal(m1[0], od1) and np.array_equal(m1[1], oe1)
            cs[0].layer_eval(layer, *m1)
            cs[1].layer_eval(layer, *m0)
        and_ids = [i for i, k in enumerate(g.kind) if k == "and"]
        for c, p in zip(cs, (p0, p1)):
            assert np.array_equal(c.values()[circ.n_inputs:], p.out[and_ids])
        assert e0.lib.sot_mt_remaining(e0.h) == n - 49400 == e1.lib.sot_mt_remaining(e1.h)
        # every LUT gate of every S-box gets its 247 subset products (LutEvalShares, aby2_lut.rs:761-781)
        shares = cs[0].lut_eval_shares()
        assert len(shares) == 200 * 8 and all(len(v) == 247 for v in shares)
    finally:
        e0.close()
        e1.close()


def _aes128_and_layer_profile():
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "aes128_bristol_and_layers.json")))
    return d["and_gates_per_layer"]


def test_aes128_bristol_and_layers_config5(fa):
    """BASELINE config 5: multiplication triples for the Baseline AES-128 Bristol circuit from the SAME device store the
    FLUTE LUT path draws on. The AND-layer profile is the real circuit's (tests/golden/aes128_bristol_and_layers.json,
    computed from Baseline/bristol/aes_128.bristol by tools/bristol_and_layers.py: 6 400 AND gates in 60 layers of
    180 / 20 / 40 / 140 / 100 / 160 per round; numbers only, the netlist is not shipped, so the wiring inside the
    profile is synthetic): k instances run as Boolean GMW layers, once layer by layer and once through
    sot_and_circ_run_layers_peer; the reconstructed outputs are the ANDs of the reconstructed inputs and the store
    shrinks by 6 400 per instance -- 7.7 times fewer triples per AES than the LUT path's 49 400."""
    prof = _aes128_and_layer_profile()
    assert sum(prof) == 6400 == fa.AES128_BRISTOL_AND_GATES and len(prof) == 60
    n = 1 << 16
    e0, e1, _ = _two_providers(fa, n, 2, 45)
    try:
        k = 5
        rng = np.random.default_rng(5)
        circ = fa.gmw_and_layers([c * k for c in prof], 256 * k, rng)
        assert circ.n_gates == 6400 * k and circ.n_layers == 60
        ins = [rng.integers(0, 2, circ.n_inputs, dtype=np.uint8) for _ in range(2)]
        cs = [fa.AndCircuit(e, circ, p) for p, e in enumerate((e0, e1))]
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        for layer in range(circ.n_layers):
            cs[0].layer_msg(layer, download=False)
            cs[1].layer_msg(layer, download=False)
            cs[0].layer_eval_from(layer, cs[1])
            cs[1].layer_eval_from(layer, cs[0])
        v = [c.values() for c in cs]
        prod = v[0] ^ v[1]
        assert np.array_equal(prod[circ.n_inputs:], prod[circ.x_src] & prod[circ.y_src])
        assert e0.lib.sot_mt_remaining(e0.h) == n - 6400 * k
        assert n // fa.AES128_BRISTOL_AND_GATES == 10 and n // fa.AES128_LUT_SETUP_MTS == 1
        # the same 60 layers in ONE call: the next 6400 k triples of each store, so other share values, the same products
        for c, i in zip(cs, ins):
            c.set_inputs(i)
        cs[0].run_layers_with(cs[1])
        w = [c.values() for c in cs]
        prod2 = w[0] ^ w[1]
        assert np.array_equal(prod2, prod)
        assert e0.lib.sot_mt_remaining(e0.h) == n - 2 * 6400 * k == e1.lib.sot_mt_remaining(e1.h)
    finally:
        e0.close()
        e1.close()


def test_run_layers_peer_and_eval_msg_equal_the_layerwise_path(fa):
    """sot_and_circ_run_layers_peer (one fused launch per layer for both parties) and sot_and_circ_layer_eval_msg (evaluate +
    next message in one call) against the oracle's two-party run: same messages, same value bits, same triple consumption
    as layer_msg / layer_eval."""
    from oracle import lut_setup_oracle as lo
    from flute_amd.lib import SotError
    luts = [tuple(l) for l in CIRCUITS["sbox8_x8"]] + [(1, 70, 3), (64, 65, 66, 67, 68)]
    n = 3000
    e0, e1, mts = _two_providers(fa, n, 2, 43)
    try:
        circ = fa.build_setup_circuit(luts)
        rng = np.random.default_rng(18)
        in0 = rng.integers(0, 2, circ.n_inputs, dtype=np.uint8)
        in1 = rng.integers(0, 2, circ.n_inputs, dtype=np.uint8)
        r = lo.run_two_party(luts, in0, in1, mts[0], mts[1])
        g, p0, p1, tr = r[0], r[2], r[3], r[4]
        and_ids = [i for i, k in enumerate(g.kind) if k == "and"]
        c0, c1 = fa.AndCircuit(e0, circ, 0), fa.AndCircuit(e1, circ, 1)
        # (a) eval + next message in one call, messages through the host
        c0.set_inputs(in0)
        c1.set_inputs(in1)
        m0, m1 = c0.layer_msg(0), c1.layer_msg(0)
        for layer in range(circ.n_layers):
            (od0, oe0), (od1, oe1), _ = tr[layer]
            assert np.array_equal(m0[0], od0) and np.array_equal(m0[1], oe0) and np.array_equal(m1[0], od1) and \
                np.array_equal(m1[1], oe1), "messages of layer %d differ from the oracle" % layer
            n0 = c0.layer_eval_msg(layer, *m1)
            n1 = c1.layer_eval_msg(layer, *m0)
            m0, m1 = n0, n1
        assert m0 is None and m1 is None
        for c, p in ((c0, p0), (c1, p1)):
            assert np.array_equal(c.values()[circ.n_inputs:], p.out[and_ids]), "eval_msg path: values differ from the oracle"
        assert e0.lib.sot_mt_remaining(e0.h) == n - circ.n_gates
        with pytest.raises(SotError):  # nothing pending any more
            c0.layer_eval_msg(0, np.zeros(8, np.uint8), np.zeros(8, np.uint8))
        # (b) all layers of both parties in one call, on fresh stores so that the triples are the oracle's again
        c0.close()
        c1.close()
    finally:
        e0.close()
        e1.close()
    e0, e1, mts = _two_providers(fa, n, 2, 43)
    try:
        c0, c1 = fa.AndCircuit(e0, circ, 0), fa.AndCircuit(e1, circ, 1)
        c0.set_inputs(in0)
        c1.set_inputs(in1)
        assert e0.lib.sot_and_circ_run_layers_peer(c0.h, c0.h, 0, 0) != 0  # one circuit twice
        c0.run_layers_with(c1, 0, 2)  # in two pieces: layers 0..2, then the rest
        c0.run_layers_with(c1, 3)
        for c, p in ((c0, p0), (c1, p1)):
            assert np.array_equal(c.values()[circ.n_inputs:], p.out[and_ids]), "run_layers_peer: values differ from the oracle"
        assert e0.lib.sot_mt_remaining(e0.h) == n - circ.n_gates == e1.lib.sot_mt_remaining(e1.h)
        with pytest.raises(SotError):
            c0.run_layers_with(c1, 0)  # already evaluated
    finally:
        e0.close()
        e1.close()
