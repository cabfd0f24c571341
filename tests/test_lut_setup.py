"""FLUTE function-dependent setup: host-side circuit builder vs the independent oracle restatement,
the reference's own unit tests restated, and the protocol property. CPU only; the GPU part is in test_gpu_lut_setup.py."""
import numpy as np
import pytest

from flute_amd.lut_setup import build_setup_circuit
from oracle import lut_setup_oracle as lo


def test_and_eval_kat():
    """crates/gmw/src/evaluate.rs:24-40 `and_eval`: p0 = [0, 1], p1 = [1, 0], zero triples -> a0 ^ a1 = 1"""
    g = lo.Graph()
    a, b = g.add("in"), g.add("in")
    t = g.add("and", [a, b])
    z = (np.zeros(1, np.uint8),) * 3
    p0, p1 = lo.Party(g, 0, [0, 1], *z), lo.Party(g, 1, [1, 0], *z)
    for p in (p0, p1):
        p.non_interactive([a, b])
    m0, m1 = p0.compute_msg([t]), p1.compute_msg([t])
    p0.evaluate([t], *m1)
    p1.evaluate([t], *m0)
    assert p0.out[t] ^ p1.out[t] == 1


def test_parent_gates_order_kat():
    """crates/gmw/src/circuit/base_circuit.rs:644-655 `parent_gates`: wires (from_0, to), (from_1, to) -> [from_1, from_0]"""
    g = lo.Graph()
    f0, f1 = g.add("in"), g.add("in")
    to = g.add("and", [f0, f1])
    assert g.parent_gates(to) == [f1, f0]


def _circuits():
    sbox = [tuple(range(8))] * 8  # sbox8.lut: 8 LUT gates over the same 8 inputs (1 grouped LUT, 8 outputs)
    rng = np.random.default_rng(3)
    mixed = [tuple(int(v) for v in rng.choice(12, size=int(k), replace=False)) for k in rng.integers(2, 6, 9)]
    overlap = [(0, 1, 2), (2, 1, 3), (0, 1, 2, 3), (5, 4)]  # shared sub-subsets must hit the cache
    batched = [((i, j) for j in range(4)) for i in range(5)]  # 5 instances of a 4-input LUT, distinct parents
    return {"sbox8": sbox, "mixed": mixed, "overlap": overlap, "batched": [tuple(b) for b in batched]}


@pytest.mark.parametrize("name", ["sbox8", "mixed", "overlap", "batched"])
def test_builder_matches_oracle(name):
    luts = _circuits()[name]
    c = build_setup_circuit(luts)
    g, circ_inputs, out_ids = lo.build(luts)
    n_in = sum(1 for k in g.kind if k == "in")
    and_ids = [i for i, k in enumerate(g.kind) if k == "and"]
    assert c.n_inputs == n_in == len(circ_inputs) and c.n_gates == len(and_ids)
    # oracle gate ids: inputs and ANDs interleave in creation order; map them to the builder's value ids
    in_ids = [i for i, k in enumerate(g.kind) if k == "in"]
    vid = {gid: j for j, gid in enumerate(in_ids)}
    vid.update({gid: n_in + j for j, gid in enumerate(and_ids)})
    for j, gid in enumerate(and_ids):
        x, y = g.parent_gates(gid)  # what compute_msg receives as (x, y)
        assert (vid[x], vid[y]) == (int(c.x_src[j]), int(c.y_src[j]))
    # layers: interactive gates per layer, sorted by id == the builder's execution order
    inter_layers = [inter for _, inter in lo.layers(g) if inter]
    assert [len(l) for l in inter_layers] == [int(v) for v in c.layer_counts]
    flat = [vid[gid] - n_in for l in inter_layers for gid in l]
    assert flat == [int(v) for v in c.exec_order]
    # outputs of every LUT gate in powerset order
    for outs_o, outs_b in zip(out_ids, c.lut_outputs):
        assert [vid[g.wired[o][0]] for o in outs_o] == list(outs_b)
    if name == "sbox8":
        assert c.n_gates == 247 and [int(v) for v in c.layer_counts] == [28, 56, 70, 56, 28, 8, 1]
        assert all(len(o) == 247 for o in c.lut_outputs)


@pytest.mark.parametrize("name", ["sbox8", "mixed", "overlap", "batched"])
def test_oracle_subset_product_property(name):
    """XOR of the parties' output shares of subset S == AND over S of the reconstructed private masks -- what
    LutEvalShares must satisfy for the online phase (aby2_lut.rs:209-277)."""
    luts = _circuits()[name]
    rng = np.random.default_rng(11)
    g, circ_inputs, out_ids = lo.build(luts)
    n_in, n_and = len(circ_inputs), sum(1 for k in g.kind if k == "and")
    in0, in1 = rng.integers(0, 2, n_in, dtype=np.uint8), rng.integers(0, 2, n_in, dtype=np.uint8)
    m0, m1 = lo.dealer_mts(n_and + 13, rng)  # 13 spare triples stay at the FRONT of the store
    _, _, p0, p1, tr = lo.run_two_party(luts, in0, in1, m0, m1)
    assert len(p0.mts[0]) == 13 and np.array_equal(p0.mts[0], m0[0][:13])
    lam = in0 ^ in1
    inputs_of = {p: i for i, p in enumerate(circ_inputs)}
    from itertools import combinations
    for parents, outs in zip(luts, out_ids):
        ins = sorted(inputs_of[p] for p in parents)
        subs = [s for k in range(2, len(ins) + 1) for s in combinations(ins, k)]
        assert len(subs) == len(outs)
        for s, o in zip(subs, outs):
            assert (p0.out[o] ^ p1.out[o]) == np.bitwise_and.reduce(lam[list(s)])


def test_replicated_template_equals_generic_builder():
    from flute_amd.lut_setup import replicate_setup_circuit
    B = 5
    tmpl = build_setup_circuit([tuple(range(8))] * 8)
    full = build_setup_circuit([tuple((i, j) for j in range(8)) for i in range(B) for _ in range(8)])
    rep = replicate_setup_circuit(tmpl, B)
    assert rep.n_inputs == full.n_inputs and rep.n_gates == full.n_gates
    for name in ("x_src", "y_src", "depth", "exec_order", "layer_counts"):
        assert np.array_equal(getattr(rep, name), getattr(full, name)), name
    assert all(np.array_equal(a, b) for a, b in zip(rep.lut_outputs, full.lut_outputs))


def test_aes128_lut_circuit_composition():
    """BASELINE config 4: AES-128 composed of 200 sbox8.lut instances (160 SubBytes + 40 key schedule). The setup
    circuit built gate by gate from the LUT gates in interactive_iter() order (LUT layer-major over the batch) equals
    the replicated sbox8 template, and costs 200 x 247 = 49 400 triples per instance in 7 rounds."""
    from flute_amd.lut_setup import (AES128_LUT_SETUP_MTS, AES128_SBOXES, aes128_lut_gate_parents,
                                     aes128_setup_circuit)
    batch = 2
    parents = aes128_lut_gate_parents(batch)
    assert len(parents) == batch * AES128_SBOXES * 8
    a = build_setup_circuit(parents)
    b = aes128_setup_circuit(batch)
    assert a.n_inputs == b.n_inputs == batch * AES128_SBOXES * 8
    assert a.n_gates == b.n_gates == batch * AES128_LUT_SETUP_MTS == batch * 49400
    assert a.n_layers == b.n_layers == 7
    assert np.array_equal(a.x_src, b.x_src) and np.array_equal(a.y_src, b.y_src) and np.array_equal(a.depth, b.depth)
    assert np.array_equal(a.layer_counts, b.layer_counts)
    assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(a.lut_outputs, b.lut_outputs))
    # one S-box of the composition against the independent oracle builder
    g, _, _ = lo.build(parents[:8])
    assert sum(1 for k in g.kind if k == "and") == 247


def test_gmw_and_layers_shape():
    from flute_amd.lut_setup import AES128_BRISTOL_AND_GATES, gmw_and_layers
    rng = np.random.default_rng(0)
    counts = [AES128_BRISTOL_AND_GATES // 40] * 40  # 6 400 ANDs in 40 layers
    c = gmw_and_layers(counts, 256, rng)
    assert c.n_gates == AES128_BRISTOL_AND_GATES and c.n_layers == 40
    xs, ys = c.exec_arrays()
    first = np.concatenate([[0], np.cumsum(c.layer_counts)[:-1]])
    for l in range(c.n_layers):  # operands come from inputs or lower layers
        sl = slice(int(first[l]), int(first[l] + c.layer_counts[l]))
        assert xs[sl].max() < 256 + first[l] and ys[sl].max() < 256 + first[l]
