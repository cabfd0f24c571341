"""Host-side mirror of FLUTE's function-dependent setup (crates/gmw/src/protocols/aby2_lut.rs:699-783):
LutSetupProvider::setup builds a Boolean AND circuit over the private mask bits of every LUT gate's inputs -- one AND
per subset of size >= 2 of the (sorted) inputs, chained through the smaller subset (setup_data_circ, :321-370) -- and
evaluates it with BooleanGmw and multiplication triples, one network round per subset size.

The graph walk stays on the host (it is the reference's circuit IR); what this module produces is
the flat description the device evaluates: per AND gate the two operand ids, per layer the gate count. The AND layers
themselves (boolean_gmw.rs:40-95, evaluate.rs:5-21) run on the GPU behind sot_and_circ_* with the triples taken from
the END of the context's device-resident MT store (Executor::execute, executor.rs:139: split_off_last per layer).

Operand order. `lhs & rhs` wires [lhs, rhs] (share_wrapper.rs:165-180) and the executor reads a gate's inputs through
parent_gates(), which yields them in REVERSE insertion order (base_circuit.rs:122-133, test :644-655); so in
and::compute_shares(x, y, mt) (boolean_gmw.rs:54-61) x = rhs and y = lhs:
   [a, b]            -> a & b           -> x = b,          y = a
   [subset.., last]  -> last & subset   -> x = subset_out, y = last
Gate order. Gate ids grow in creation order; a layer's interactive gates are sorted by id (base_circuit.rs:481-484);
an AND over a subset of size k sits in layer k - 2 (its parents are an input and the AND of size k - 1).
"""
import ctypes as C

import numpy as np

from .lib import check


class SetupCircuit:
    """Flat form of the setup circuit. Value ids: 0 .. n_inputs-1 are the private-share input bits in the order setup()
    pushes them to circ_inputs (first use, aby2_lut.rs:725-741); n_inputs + i is AND gate i in CREATION order."""

    def __init__(self, n_inputs, x_src, y_src, depth, lut_outputs):
        self.n_inputs = n_inputs
        self.x_src = np.asarray(x_src, dtype=np.uint32)
        self.y_src = np.asarray(y_src, dtype=np.uint32)
        self.depth = np.asarray(depth, dtype=np.uint32)  # layer of every gate (0 = subsets of size 2)
        self.lut_outputs = lut_outputs  # per LUT gate: value ids of its 2^p - p - 1 subset products, powerset order
        self.n_gates = len(self.x_src)
        self.n_layers = int(self.depth.max()) + 1 if self.n_gates else 0
        # execution order = (layer, gate id); value ids are renumbered so that a layer's outputs are contiguous
        order = np.lexsort((np.arange(self.n_gates), self.depth))
        self.exec_order = order
        self.layer_counts = np.bincount(self.depth, minlength=self.n_layers).astype(np.uint32)
        rank = np.empty(self.n_gates, dtype=np.int64)
        rank[order] = np.arange(self.n_gates)
        self._rank = rank

    def exec_id(self, value_id):
        """creation-order value id -> execution-order value id (inputs unchanged)"""
        v = np.asarray(value_id, dtype=np.int64)
        return np.where(v < self.n_inputs, v, self.n_inputs + self._rank[np.maximum(v - self.n_inputs, 0)])

    def exec_arrays(self):
        xs = self.exec_id(self.x_src[self.exec_order]).astype(np.uint32)
        ys = self.exec_id(self.y_src[self.exec_order]).astype(np.uint32)
        return xs, ys


def build_setup_circuit(lut_gate_parents):
    """lut_gate_parents: for every interactive LUT gate, in circuit.interactive_iter() order, the ids of its parent
    gates (any hashable, e.g. (circuit_id, gate_id)). Follows setup() :717-741 (inputs numbered at first use, the
    gate's input shares SORTED -- ShareWrapper orders by its output gate id, i.e. by input number) and setup_data_circ
    :321-358 (powerset without the empty and single-element sets, cache keyed by the subset)."""
    from itertools import combinations
    input_of = {}
    x_src, y_src, depth = [], [], []
    cache = {}
    lut_outputs = []
    lut_gate_parents = [tuple(p) for p in lut_gate_parents]
    n_inputs_total = len({p for parents in lut_gate_parents for p in parents})
    last_parents, last_outs = None, None
    for parents in lut_gate_parents:
        if parents == last_parents:  # same input set as the previous LUT gate: every subset is a cache hit
            lut_outputs.append(list(last_outs))
            continue
        ins = []
        for p in parents:
            if p not in input_of:
                input_of[p] = len(input_of)
            ins.append(input_of[p])
        ins.sort()
        outs = []
        for k in range(2, len(ins) + 1):
            for sub in combinations(ins, k):
                if sub not in cache:
                    gid = n_inputs_total + len(x_src)
                    if k == 2:
                        x_src.append(sub[1]), y_src.append(sub[0]), depth.append(0)
                    else:
                        x_src.append(cache[sub[:-1]]), y_src.append(sub[-1]), depth.append(k - 2)
                    cache[sub] = gid
                outs.append(cache[sub])
        lut_outputs.append(outs)
        last_parents, last_outs = parents, outs
    return SetupCircuit(n_inputs_total, x_src, y_src, depth, lut_outputs)


def replicate_setup_circuit(template, batch):
    """The setup circuit of `batch` sub-circuit instances of a ONE-LAYER LUT circuit (e.g. to_eval/sbox8.lut, #DEPTH 1),
    as flute_benchmark builds it (bench.rs:367-384): interactive_iter() then visits instance 0's LUT gates, instance
    1's, ... so the creation order is instance-major and build_setup_circuit(parents of all instances) equals the
    template with its ids shifted per instance -- built here with numpy instead of a Python loop per gate.
    (A LUT circuit with several layers is visited layer-major; use build_setup_circuit for it.)"""
    t = template
    ni, ng = t.n_inputs, t.n_gates
    inst = np.arange(batch, dtype=np.int64)[:, None]

    def shift(src):
        src = src.astype(np.int64)[None, :]
        return np.where(src < ni, src + inst * ni, batch * ni + (src - ni) + inst * ng).reshape(-1)

    outs = []
    if t.lut_outputs:
        tmpl = [np.asarray(o, dtype=np.int64) for o in t.lut_outputs]
        if len({len(o) for o in tmpl}) == 1:  # same fan-in everywhere: one broadcast, rows are views
            arr = (batch * ni + (np.stack(tmpl) - ni))[None, :, :] + (inst * ng)[:, :, None]
            outs = list(arr.reshape(-1, arr.shape[-1]))
        else:
            for i in range(batch):
                outs.extend(batch * ni + (o - ni) + i * ng for o in tmpl)
    return SetupCircuit(batch * ni, shift(t.x_src), shift(t.y_src), np.tile(t.depth, batch), outs)


# ---- BASELINE configs 4 and 5: AES-128 ---------------------------------------------------------------------------------
# to_eval/ holds no AES-128 LUT circuit (only sbox8.lut and operators/*), so the "full AES-128 LUT circuit" is composed
# from sbox8.lut instances: FIPS-197 AES-128 has 10 rounds x 16 SubBytes S-boxes and a key schedule of 10 x 4 S-boxes;
# with a secret-shared key the schedule runs in-circuit (the Baseline Bristol circuit does the same: its 6 400 AND gates
# are 200 S-boxes x 32 ANDs). Everything else in AES (ShiftRows, MixColumns, AddRoundKey, Rcon) is XOR-linear: free in
# FLUTE's setup, which only needs the subset products of every LUT gate's input masks (aby2_lut.rs:699-783).
AES128_SBOXES = 160 + 40
SBOX8_SETUP_ANDS = 2 ** 8 - 8 - 1  # 247 per 8-input grouped LUT (its 8 outputs share the subsets through the cache)
AES128_LUT_SETUP_MTS = AES128_SBOXES * SBOX8_SETUP_ANDS  # 49 400 multiplication triples per AES-128 instance
# crates/flute_benchmark/Baseline/bristol/aes_128.bristol: header ":1-2" "36663 36919" (gates, wires); 6 400 of the gate
# lines end in AND (counted in the survey container; the file itself is not copied)
AES128_BRISTOL_AND_GATES = 6400


def aes128_lut_gate_parents(batch):
    """LUT gates of `batch` AES-128 sub-circuit instances in circuit.interactive_iter() order (bench.rs:367-384 builds
    the batch as identical sub-circuits; the layer iterator visits LUT layer by LUT layer, inside a layer sub-circuit by
    sub-circuit, then gate id): LUT layer r - 1 holds round r's 16 SubBytes S-boxes and 4 key-schedule S-boxes of every
    instance. An S-box is sbox8.lut: 8 LUT gates (one per output bit) over the same 8 parents. Parent ids are
    (instance, round, sbox, input bit): the wires into different S-boxes are different wires."""
    parents = []
    for rnd in range(10):
        for inst in range(batch):
            for sb in range(20):  # 0..15 SubBytes, 16..19 key schedule
                p = tuple((inst, rnd, sb, bit) for bit in range(8))
                parents.extend([p] * 8)
    return parents


def aes128_setup_circuit(batch):
    """setup circuit of `batch` AES-128 instances: every S-box has its own 8 inputs, so it equals the sbox8 template
    replicated 200 * batch times with unit u = (round * batch + instance) * 20 + sbox (tests/test_lut_setup.py checks this
    against build_setup_circuit(aes128_lut_gate_parents(batch)))"""
    return replicate_setup_circuit(build_setup_circuit([tuple(range(8))] * 8), AES128_SBOXES * batch)


def gmw_and_layers(layer_counts, n_inputs, rng):
    """A Boolean GMW AND workload in the flat form of sot_and_circ_create: layer l has layer_counts[l] AND gates whose
    operands are random earlier values (inputs or gates of lower layers). Stands in for the AND layers of a Bristol
    circuit whose netlist is not shipped (Baseline/bristol/aes_128.bristol: 6 400 AND gates); what it measures and tests
    is the per-layer triple consumption and evaluation, which depends on the counts only."""
    x_src, y_src, depth = [], [], []
    avail = n_inputs
    for layer, cnt in enumerate(layer_counts):
        x_src.append(rng.integers(0, avail, cnt))
        y_src.append(rng.integers(0, avail, cnt))
        depth.append(np.full(cnt, layer))
        avail += cnt
    return SetupCircuit(n_inputs, np.concatenate(x_src), np.concatenate(y_src), np.concatenate(depth), [])


class AndCircuit:
    """Device evaluator of a SetupCircuit on an Engine (C ABI sot_and_circ_*). Mirrors Executor<BooleanGmw>:
    per layer compute_msg -> exchange -> evaluate_interactive; MTs come from the end of the engine's MT store."""

    def __init__(self, engine, circ, party_id):
        self.engine, self.circ, self.party_id = engine, circ, party_id
        xs, ys = circ.exec_arrays()
        h = C.c_void_p()
        lc = np.ascontiguousarray(circ.layer_counts, dtype=np.uint32)
        check(engine.lib.sot_and_circ_create(engine.h, circ.n_inputs, circ.n_gates, circ.n_layers,
                                             lc.ctypes.data_as(C.c_void_p), xs.ctypes.data_as(C.c_void_p),
                                             ys.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h
        engine._circuits.append(self)  # Engine.close() destroys its circuits before the context

    def close(self):
        if getattr(self, "h", None):
            self.engine.lib.sot_and_circ_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def set_inputs(self, bits):
        """bits: the private share of every circuit input (circ_inputs of setup()), one 0/1 per input"""
        b = np.asarray(bits, dtype=np.uint8)
        assert b.shape == (self.circ.n_inputs,)
        w = np.packbits(np.concatenate([b, np.zeros((-len(b)) % 64, np.uint8)]), bitorder="little")
        check(self.engine.lib.sot_and_circ_set_inputs(self.h, w.ctypes.data_as(C.c_void_p)))

    def layer_msg(self, layer, download=True):
        """BooleanGmw::compute_msg for one layer -> (d, e) as the bytes of BitVec<u8, Lsb0>::into_vec()"""
        nb = (int(self.circ.layer_counts[layer]) + 7) // 8
        if not download:
            check(self.engine.lib.sot_and_circ_layer_msg(self.h, layer, None, None))
            return None
        d = np.zeros(nb, dtype=np.uint8)
        e = np.zeros(nb, dtype=np.uint8)
        check(self.engine.lib.sot_and_circ_layer_msg(self.h, layer, d.ctypes.data_as(C.c_void_p),
                                                     e.ctypes.data_as(C.c_void_p)))
        return d, e

    def layer_eval(self, layer, peer_d, peer_e):
        """BooleanGmw::evaluate_interactive with the peer's message"""
        pd = np.ascontiguousarray(peer_d, dtype=np.uint8)
        pe = np.ascontiguousarray(peer_e, dtype=np.uint8)
        check(self.engine.lib.sot_and_circ_layer_eval(self.h, layer, self.party_id, pd.ctypes.data_as(C.c_void_p),
                                                      pe.ctypes.data_as(C.c_void_p)))

    def layer_eval_from(self, layer, peer):
        """same, the peer's message taken device-to-device from another AndCircuit on the same GPU (benchmarks)"""
        check(self.engine.lib.sot_and_circ_layer_eval_peer(self.h, layer, self.party_id, peer.h))

    def layer_eval_msg(self, layer, peer_d, peer_e, download=True):
        """evaluate_interactive of `layer` + compute_msg of layer + 1 in one call (sot_and_circ_layer_eval_msg); returns the
        next layer's (d, e), or None after the last layer"""
        pd = np.ascontiguousarray(peer_d, dtype=np.uint8)
        pe = np.ascontiguousarray(peer_e, dtype=np.uint8)
        more = layer + 1 < self.circ.n_layers
        if more and download:
            nb = (int(self.circ.layer_counts[layer + 1]) + 7) // 8
            d, e = np.zeros(nb, dtype=np.uint8), np.zeros(nb, dtype=np.uint8)
            dp, ep = d.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p)
        else:
            d = e = dp = ep = None
        check(self.engine.lib.sot_and_circ_layer_eval_msg(self.h, layer, self.party_id, pd.ctypes.data_as(C.c_void_p),
                                                          pe.ctypes.data_as(C.c_void_p), dp, ep))
        return (d, e) if more and download else None

    def run_layers_with(self, peer, first=0, last=None):
        """both parties' layers first .. last in one call (self = party 0, peer = party 1, same GPU):
        sot_and_circ_run_layers_peer -- one fused launch per layer, one synchronisation"""
        assert self.party_id == 0 and peer.party_id == 1
        last = self.circ.n_layers - 1 if last is None else last
        check(self.engine.lib.sot_and_circ_run_layers_peer(self.h, peer.h, first, last))

    def values(self):
        """all value bits (inputs, then AND outputs) in CREATION order, one 0/1 per value"""
        n = self.circ.n_inputs + self.circ.n_gates
        w = np.zeros((n + 63) // 64 + self.circ.n_layers + 1, dtype=np.uint64)
        bits = np.zeros(n, dtype=np.uint8)
        check(self.engine.lib.sot_and_circ_values(self.h, bits.ctypes.data_as(C.c_void_p)))
        out = np.empty(n, dtype=np.uint8)
        out[:self.circ.n_inputs] = bits[:self.circ.n_inputs]
        out[self.circ.n_inputs + self.circ.exec_order] = bits[self.circ.n_inputs:]
        del w
        return out

    def lut_eval_shares(self):
        """LutEvalShares of every LUT gate (aby2_lut.rs:761-781): its subset-product shares in powerset order"""
        v = self.values()
        return [v[np.asarray(o, dtype=np.int64)] for o in self.circ.lut_outputs]
