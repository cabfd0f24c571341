"""Host-side mirror of FLUTE's ONLINE LUT evaluation (crates/gmw/src/protocols/aby2_lut.rs:125-165 compute_msg /
evaluate_interactive, :209-277 LutGate::compute_delta_share, :947-964 expand) for `batch` SIMD instances of a template of
LUT gates (flute_benchmark batches sub-circuits, bench.rs:367-384). The subset products of the input lambdas come from
the setup phase (flute_amd/lut_setup.py: an evaluated AndCircuit of the batch-replicated setup circuit, resident in HBM);
this module builds the per-gate tables the device kernel needs and calls sot_lut_online_eval.

Conventions (the reference's): a LUT with p inputs has a 2^p-entry table indexed by the row j whose bit p-1-i is input i
(first input = most significant bit, parse/lut_circuit.rs:553-571 and aby2_lut.rs:959); `table` below is that list
(table[j] = output for row j), e.g. oracle-free: the Msb0 bits of the big-endian hex number of a .lut file.
The reference's own fold is all-zero (bitvec_and_fold, :1177-1188, tests #[ignore]d): this evaluates the protocol as
specified; DESIGN.md section 10."""
import ctypes as C
from itertools import combinations

import numpy as np

from .lib import check
from .lut_setup import build_setup_circuit, replicate_setup_circuit


class LutTemplate:
    """luts: [(parents, table)] in circuit.interactive_iter() order; parents = hashable wire ids in GATE INPUT order"""

    def __init__(self, luts):
        self.parents = [tuple(p) for p, _ in luts]
        self.tables = [list(t) for _, t in luts]
        for p, t in zip(self.parents, self.tables):
            if not 1 <= len(p) <= 8 or len(set(p)) != len(p) or len(t) != 1 << len(p):
                raise ValueError("LUT gates take 1 .. 8 distinct inputs and a table of 2^p entries")
        self.setup = build_setup_circuit(self.parents)  # the template's subset-product circuit
        s = self.setup
        self.n_luts = len(luts)
        # setup input number of every wire: first use over the gates, as setup() numbers them (aby2_lut.rs:725-741)
        input_of = {}
        for p in self.parents:
            for w in p:
                input_of.setdefault(w, len(input_of))
        self.input_of = input_of
        # rank of every template AND gate inside its setup layer (execution order inside a layer = gate id)
        rank = np.zeros(s.n_gates, dtype=np.int64)
        seen = np.zeros(max(s.n_layers, 1), dtype=np.int64)
        for gi in range(s.n_gates):
            rank[gi] = seen[s.depth[gi]]
            seen[s.depth[gi]] += 1
        self.tcount = np.ascontiguousarray(s.layer_counts, dtype=np.uint32) if s.n_layers else np.zeros(1, np.uint32)
        self.arity = np.array([len(p) for p in self.parents], dtype=np.uint32)
        self.tt = np.zeros((self.n_luts, 8), dtype=np.uint32)
        self.lsrc = np.zeros((self.n_luts, 256), dtype=np.uint32)
        for g, (par, tab, outs) in enumerate(zip(self.parents, self.tables, s.lut_outputs)):
            p = len(par)
            for j, bit in enumerate(tab):
                if bit:
                    self.tt[g, j >> 5] |= np.uint32(1 << (j & 31))
            ins = [input_of[w] for w in par]
            ins_sorted = sorted(ins)
            prod = {}  # sorted tuple of input numbers -> value id (creation order) of the subset product
            it = iter(outs)
            for k in range(2, p + 1):
                for sub in combinations(ins_sorted, k):
                    prod[sub] = int(next(it))
            for sm in range(1, 1 << p):
                members = sorted(ins[i] for i in range(p) if (sm >> (p - 1 - i)) & 1)
                if len(members) == 1:
                    self.lsrc[g, sm] = members[0]
                else:
                    gi = prod[tuple(members)] - s.n_inputs
                    self.lsrc[g, sm] = ((int(s.depth[gi]) + 1) << 24) | int(rank[gi])

    def setup_circuit(self, batch):
        """the setup circuit of `batch` instances (instance-major ids, as flute_benchmark's batching creates them)"""
        return replicate_setup_circuit(self.setup, batch)

    def rows(self, masked_bits):
        """public masked input bits [batch, n_wires] (column = setup input number) -> row index of every gate [batch, n_luts]"""
        mb = np.asarray(masked_bits, dtype=np.uint8)
        out = np.zeros((mb.shape[0], self.n_luts), dtype=np.uint8)
        for g, par in enumerate(self.parents):
            p = len(par)
            for i, w in enumerate(par):
                out[:, g] |= (mb[:, self.input_of[w]] & 1) << (p - 1 - i)
        return out

    def plain(self, value_bits):
        """plaintext evaluation (lut_circuit.rs:553-571) of every gate: value_bits [batch, n_wires] -> [batch, n_luts]"""
        r = self.rows(value_bits)
        return np.stack([np.asarray(self.tables[g], dtype=np.uint8)[r[:, g]] for g in range(self.n_luts)], axis=1)


def online_delta_shares(and_circuit, template, batch, masked_rows, lam_out, party_id, first_instance=0):
    """LutGate::compute_delta_share for every (instance, gate): this party's delta share bits [batch, n_luts]; the public
    output Delta is the XOR of the two parties' arrays (evaluate_interactive, aby2_lut.rs:145-165).
    and_circuit: the party's AndCircuit of template.setup_circuit(N) with every layer evaluated, of which instances
    first_instance .. first_instance + batch - 1 are evaluated here (N = batch by default); masked_rows: the
    public masked inputs as row indices (template.rows); lam_out: its shares of the output lambdas [batch, n_luts]."""
    t = template
    m = np.ascontiguousarray(masked_rows, dtype=np.uint8).reshape(batch, t.n_luts)
    lo = np.ascontiguousarray(lam_out, dtype=np.uint8).reshape(batch, t.n_luts)
    out = np.zeros((batch, t.n_luts), dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(and_circuit.engine.lib.sot_lut_online_eval(and_circuit.h, first_instance, batch, t.n_luts, t.setup.n_inputs, p(t.arity), p(t.tt),
                                                     p(t.lsrc), p(t.tcount), p(m), p(lo), party_id, p(out)))
    return out


def online_delta_shares_dev(and_circuit, template, batch, masked_rows_dev, lam_out_dev, out_dev, party_id, first_instance=0):
    """online_delta_shares with the per-item arrays as DEVICE pointers (batch x n_luts bytes each, e.g. torch uint8 tensors'
    data_ptr()) on the circuit's GPU: sot_lut_online_eval_dev. Queued on the engine's stream, not waited for
    (and_circuit.engine.sync())."""
    t = template
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    check(and_circuit.engine.lib.sot_lut_online_eval_dev(and_circuit.h, first_instance, batch, t.n_luts, t.setup.n_inputs,
                                                         p(t.arity), p(t.tt), p(t.lsrc), p(t.tcount), C.c_void_p(masked_rows_dev),
                                                         C.c_void_p(lam_out_dev), party_id, C.c_void_p(out_dev)))
