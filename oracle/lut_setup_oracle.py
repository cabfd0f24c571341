"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by flute_amd/): CPU restatement of FLUTE's function-dependent
setup, written independently of flute_amd/lut_setup.py so that the two can be compared.

Follows, line by line:
  * LutSetupProvider::setup            crates/gmw/src/protocols/aby2_lut.rs:699-783
  * LutGate::setup_data_circ           aby2_lut.rs:321-370   (itertools powerset().skip(inputs + 1), subset cache)
  * ShareWrapper & ShareWrapper        crates/gmw/src/share_wrapper.rs:165-180  (add_wired_gate(And, [lhs, rhs]))
  * BaseCircuit::parent_gates          crates/gmw/src/circuit/base_circuit.rs:122-133 (+ test :644-655: REVERSE of the
                                       wiring order -- petgraph neighbors_directed(Incoming))
  * BaseLayerIter::next                base_circuit.rs:431-494 (a gate joins the layer in which its last parent was
                                       visited; interactive gates of a layer sorted by gate id)
  * Executor::execute                  crates/gmw/src/executor.rs:74-193 (split_off_last(layer count), compute_msg,
                                       exchange, evaluate_interactive)
  * BooleanGmw::compute_msg / evaluate_interactive   protocols/boolean_gmw.rs:40-95
  * and::compute_shares / and::evaluate              crates/gmw/src/evaluate.rs:5-21
  * MulTriples::split_off_last / iter                mul_triple/mod.rs:172-185, :92-99

Parity status: the Rust crates cannot be built here, so the gate ORDER conventions above are pinned only by
reading the source and by the reference's own tests restated in tests/test_lut_setup.py (evaluate.rs:24-40 `and_eval`,
base_circuit.rs:644-655 `parent_gates`); the VALUES are pinned by the protocol property (XOR of the two parties' output
shares = AND of the reconstructed inputs). "parity unpinned" by any reference-produced vector.
"""
from itertools import chain, combinations

import numpy as np


class Graph:
    """just enough of BaseCircuit: gates with ordered incoming wires"""

    def __init__(self):
        self.kind = []  # "in" | "and" | "out"
        self.wired = []  # parents in WIRING order

    def add(self, kind, parents=()):
        self.kind.append(kind)
        self.wired.append(list(parents))
        return len(self.kind) - 1

    def parent_gates(self, g):  # base_circuit.rs:122-133
        return list(reversed(self.wired[g]))


def powerset(xs):  # itertools (Rust) powerset: by size, each size in lexicographic order of positions
    return chain.from_iterable(combinations(xs, k) for k in range(len(xs) + 1))


def build(lut_gate_parents):
    """setup() :717-752. Returns (graph, circ_input_parents, setup_outputs) -- setup_outputs[i] = the Output gate ids of
    LUT gate i in powerset order."""
    g = Graph()
    input_sw_map = {}
    circ_input_parents = []
    cache = {}
    setup_outputs = []
    for parents in lut_gate_parents:
        shares = []
        for p in parents:
            if p not in input_sw_map:  # Entry::Vacant
                input_sw_map[p] = g.add("in")
                circ_input_parents.append(p)
            shares.append(input_sw_map[p])
        shares.sort()  # gate_input_shares.sort(): ShareWrapper orders by (circuit id, output gate)
        n = len(shares)
        outs = []
        for sub in list(powerset(shares))[n + 1:]:  # .powerset().skip(inputs + 1)
            if sub not in cache:
                if len(sub) == 2:
                    a, b = sub
                    cache[sub] = g.add("and", [a, b])  # a.clone() & b
                else:
                    last = sub[-1]
                    subset_out = cache[sub[:-1]]
                    cache[sub] = g.add("and", [last, subset_out])  # last.clone() & subset_out
            outs.append(cache[sub])
        setup_outputs.append(outs)
    # .output() of every share, in LUT gate order (:743-751); the same AND may get several Output gates
    out_ids = [[g.add("out", [a]) for a in outs] for outs in setup_outputs]
    return g, circ_input_parents, out_ids


def layers(g):
    """BaseLayerIter: list of (non_interactive ids in visit order, interactive ids sorted)"""
    n = len(g.kind)
    children = [[] for _ in range(n)]
    for c in range(n):
        for p in g.wired[c]:
            children[p].append(c)
    visited = [False] * n
    next_layer = [i for i in range(n) if g.kind[i] == "in"]  # BaseLayerIter::new seeds the gates without parents
    res = []
    while True:
        to_visit, next_layer = next_layer, []
        non_int, inter = [], set()
        qi = 0
        while qi < len(to_visit):
            node = to_visit[qi]
            qi += 1
            if visited[node]:
                continue
            if g.kind[node] == "and":
                inter.add(node)
                for ch in children[node]:
                    if all(visited[b] or b in inter for b in g.wired[ch]):
                        next_layer.append(ch)
            else:
                visited[node] = True
                non_int.append(node)
                for ch in children[node]:
                    if all(visited[b] for b in g.wired[ch]):
                        to_visit.append(ch)
                    elif all(visited[b] or b in inter for b in g.wired[ch]):
                        next_layer.append(ch)
        for i in inter:
            visited[i] = True
        if not non_int and not inter:
            return res
        res.append((non_int, sorted(inter)))


class Party:
    """Executor<BooleanGmw> of one party over the setup circuit, driven layer by layer by the test"""

    def __init__(self, g, party_id, inputs, mt_a, mt_b, mt_c):
        self.g, self.party_id = g, party_id
        self.out = np.zeros(len(g.kind), dtype=np.uint8)
        self.inputs = list(inputs)
        self.mts = [np.asarray(mt_a, np.uint8).copy(), np.asarray(mt_b, np.uint8).copy(),
                    np.asarray(mt_c, np.uint8).copy()]
        self.input_gates = [i for i, k in enumerate(g.kind) if k == "in"]

    def split_off_last(self, k):  # mul_triple/mod.rs:172-185
        n = len(self.mts[0])
        assert k <= n
        tail = [m[n - k:] for m in self.mts]
        self.mts = [m[:n - k] for m in self.mts]
        return tail

    def non_interactive(self, ids):
        for i in ids:
            if self.g.kind[i] == "in":
                self.out[i] = self.inputs[self.input_gates.index(i)]
            else:  # Output: forwards its single input
                self.out[i] = self.out[self.g.parent_gates(i)[0]]

    def compute_msg(self, ids):  # boolean_gmw.rs:40-67
        self.layer_mts = self.split_off_last(len(ids))
        a, b, _ = self.layer_mts
        d = np.zeros(len(ids), np.uint8)
        e = np.zeros(len(ids), np.uint8)
        for j, gid in enumerate(ids):
            x, y = (self.out[p] for p in self.g.parent_gates(gid))
            d[j] = x ^ a[j]  # and::compute_shares
            e[j] = y ^ b[j]
        self.own = (d, e)
        return np.packbits(d, bitorder="little"), np.packbits(e, bitorder="little")  # BitVec<u8, Lsb0>::into_vec

    def evaluate(self, ids, peer_d, peer_e):  # boolean_gmw.rs:69-95, evaluate.rs:12-21
        a, b, c = self.layer_mts
        pd = np.unpackbits(np.asarray(peer_d, np.uint8), bitorder="little")
        pe = np.unpackbits(np.asarray(peer_e, np.uint8), bitorder="little")
        for j, gid in enumerate(ids):
            d = self.own[0][j] ^ pd[j]
            e = self.own[1][j] ^ pe[j]
            res = (d & b[j]) ^ (e & a[j]) ^ c[j]
            if self.party_id == 0:
                res ^= d & e
            self.out[gid] = res


def run_two_party(lut_gate_parents, inputs0, inputs1, mts0, mts1):
    """both parties in one process; returns (graph, out_ids, party0, party1, transcript) with transcript[l] =
    ((d0, e0), (d1, e1)) message bytes of interactive layer l"""
    g, _, out_ids = build(lut_gate_parents)
    p0 = Party(g, 0, inputs0, *mts0)
    p1 = Party(g, 1, inputs1, *mts1)
    transcript = []
    for non_int, inter in layers(g):
        p0.non_interactive(non_int)
        p1.non_interactive(non_int)
        if not inter:
            continue
        m0 = p0.compute_msg(inter)
        m1 = p1.compute_msg(inter)
        p0.evaluate(inter, *m1)
        p1.evaluate(inter, *m0)
        transcript.append((m0, m1, inter))
    return g, out_ids, p0, p1, transcript


def dealer_mts(n, rng):
    """trusted-dealer triples for the CPU-only tests: c0 ^ c1 = (a0 ^ a1) & (b0 ^ b1)"""
    a0, a1, b0, b1, c0 = (rng.integers(0, 2, n, dtype=np.uint8) for _ in range(5))
    c1 = ((a0 ^ a1) & (b0 ^ b1)) ^ c0
    return (a0, b0, c0), (a1, b1, c1)
