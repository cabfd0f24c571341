"""TEST INFRASTRUCTURE ONLY (imported by tests/, never by flute_amd/): CPU restatement of FLUTE's ONLINE LUT evaluation
(SURVEY 8f N2), one LUT gate and one instance at a time, in plain Python.

Follows crates/gmw/src/protocols/aby2_lut.rs:
  * LutGate::compute_delta_share   :209-277  (powerset of the expanded input vectors, reversed; eval shares popped in
                                              the order input lambdas, then the setup's subset products)
  * expand                         :947-964  (x_i[j] = !m_i ^ bit (p - 1 - i) of row j, rows j with output_mask[j] set)
  * evaluate_interactive           :145-165  (public output = XOR of the two parties' delta shares)
  * plaintext LUT semantics        parse/lut_circuit.rs:553-571 (out = unexpanded[inp.load_be()], Msb0 table index),
                                   :355-399 (the hex table is a big-endian number, bit 0 of the table = its MSB)

ONE DELIBERATE DEVIATION, the reason this row is "parity unpinned": the reference's bitvec_and_fold (:1177-1188) starts its
AND-fold from BitVec::repeat(false, size), so every fold is all-zero and compute_delta_share returns the output share's
private bit -- it does not compute the FLUTE delta (its integration tests are #[ignore = "issue #6"],
gmw/tests/boolean_aby2.rs:15). and_fold() below starts from all-ones, which is what the protocol needs; everything else is
the reference's control flow. What pins the result instead: reconstructing the two parties' outputs gives the plaintext
LUT value for every input (tests/test_lut_eval.py), and to_eval/sbox8.lut's tables are the FIPS-197 S-box.
"""
from itertools import chain, combinations


def powerset(xs):  # itertools (Rust) powerset: by size, each size in lexicographic order of positions
    xs = list(xs)
    return list(chain.from_iterable(combinations(xs, k) for k in range(len(xs) + 1)))


def table_bits(table_hex, p):
    """output_mask as the parser builds it (lut_circuit.rs:355-399): Msb0 bits of the big-endian number, 2^p of them"""
    n = 1 << p
    v = int(table_hex, 16)
    return [(v >> (n - 1 - j)) & 1 for j in range(n)]


def plain_eval(table, inputs):
    """lut_circuit.rs:553-571: inputs in wire order, first wire = most significant bit of the row index"""
    idx = 0
    for b in inputs:
        idx = (idx << 1) | (b & 1)
    return table[idx]


def expand(p, table, masks):
    """aby2_lut.rs:947-964 -> p vectors over the set rows of the table"""
    out = [[] for _ in range(p)]
    for i in range(p):
        for j, bit in enumerate(table):
            if not bit:
                continue
            tt = (j >> (p - i - 1)) & 1  # j.view_bits::<Lsb0>()[p - i - 1]
            out[i].append((1 ^ masks[i]) ^ tt)
    return out


def and_fold(size, vecs):
    acc = [1] * size  # the reference starts from 0 here (see the module docstring)
    for v in vecs:
        acc = [a & b for a, b in zip(acc, v)]
    return acc


def compute_delta_share(party_id, table, in_public, in_private, eval_shares, out_private):
    """aby2_lut.rs:209-277. in_public / in_private: the masked values m_i and this party's lambda shares of the p inputs
    (gate input order); eval_shares: this party's shares of the subset products of size >= 2 in powerset order (what
    setup() returns for the gate, LutEvalShares::shares); out_private: its share of the output lambda."""
    p = len(in_public)
    x_vecs = expand(p, table, in_public)
    shares = list(in_private) + list(eval_shares)
    shares.reverse()
    pset = powerset(range(p))
    whole = pset.pop()
    pset.reverse()
    nrows = len(x_vecs[0]) if p else 0
    red = 0
    for q in pset:
        and_x_q = and_fold(nrows, [x_vecs[i] for i in q])
        if shares:
            es = shares.pop()
            r = 0
            for bit in and_x_q:
                r ^= bit & es
            red ^= r
        else:
            assert not q, "a missing eval share is only allowed for the empty set"
    if party_id == 1:
        wm = 0
        for bit in and_fold(nrows, [x_vecs[i] for i in whole]):
            wm ^= bit
        return wm ^ red ^ out_private
    return red ^ out_private


def lambda_products(lam):
    """plaintext subset products of the lambda bits (size >= 2, powerset order): what the setup phase shares"""
    out = []
    for sub in powerset(range(len(lam)))[len(lam) + 1:]:
        v = 1
        for i in sub:
            v &= lam[i]
        out.append(v)
    return out
