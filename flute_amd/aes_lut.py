"""AES-128 as a FLUTE LUT circuit (BASELINE config 4): 200 instances of to_eval/sbox8.lut -- 16 SubBytes S-boxes per
round and the 4 SubWord S-boxes of the key schedule, 10 rounds -- joined by XOR-linear layers (ShiftRows, MixColumns,
AddRoundKey, RotWord, Rcon), evaluated on ABY2.0-style Delta shares as the reference's Aby2Lut protocol does
(crates/gmw/src/protocols/aby2_lut.rs: Share {public: Delta = value ^ lambda, private: this party's share of lambda};
XOR / NOT / constants are local, :279-319; LUT gates take one setup phase for the whole circuit, :699-783, and one
message round per LUT layer, :125-165).

The linear layers are host code here as they are host graph code in the reference; the two interactive parts run on the
GPU: the setup AND layers of all 200 * batch S-boxes at once (flute_amd/lut_setup.py) and, per AES round, the online
evaluation of that round's 20 * batch S-boxes (flute_amd/lut_eval.py). to_eval holds no AES-128 LUT file; the wiring
below is FIPS-197's, the S-box is the reference's sbox8.lut, whose table under the reference's row convention maps a row
index idx to S[255 - idx] (tests/test_lut_eval.py), i.e. its input wires are addr[i] = NOT bit (7 - i) of the byte.

Bits of a byte are stored least-significant first: arrays [..., 8] of 0/1."""
import numpy as np

from .lut_eval import LutTemplate, online_delta_shares

ROUNDS = 10
SBOXES_PER_ROUND = 20  # 16 SubBytes + 4 SubWord
RCON = [0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36]


def bytes_to_bits(b):
    return np.unpackbits(np.asarray(b, dtype=np.uint8)[..., None], axis=-1, bitorder="little")


def bits_to_bytes(bits):
    return np.packbits(np.asarray(bits, dtype=np.uint8), axis=-1, bitorder="little")[..., 0]


def _xtime(b):
    o = np.empty_like(b)
    o[..., 0] = b[..., 7]
    o[..., 1] = b[..., 0] ^ b[..., 7]
    o[..., 2] = b[..., 1]
    o[..., 3] = b[..., 2] ^ b[..., 7]
    o[..., 4] = b[..., 3] ^ b[..., 7]
    o[..., 5] = b[..., 4]
    o[..., 6] = b[..., 5]
    o[..., 7] = b[..., 6]
    return o


def _shift_rows(s):  # s: [batch, 16, 8], byte 4c + r = row r, column c
    o = np.empty_like(s)
    for c in range(4):
        for r in range(4):
            o[:, 4 * c + r] = s[:, 4 * ((c + r) % 4) + r]
    return o


def _mix_columns(s):
    o = np.empty_like(s)
    for c in range(4):
        a = [s[:, 4 * c + r] for r in range(4)]
        x = [_xtime(v) for v in a]
        o[:, 4 * c + 0] = x[0] ^ (x[1] ^ a[1]) ^ a[2] ^ a[3]
        o[:, 4 * c + 1] = a[0] ^ x[1] ^ (x[2] ^ a[2]) ^ a[3]
        o[:, 4 * c + 2] = a[0] ^ a[1] ^ x[2] ^ (x[3] ^ a[3])
        o[:, 4 * c + 3] = (x[0] ^ a[0]) ^ a[1] ^ a[2] ^ x[3]
    return o


def aes128_walk(pt_bits, key_bits, sbox, constants):
    """FIPS-197 AES-128 on bit arrays [batch, 16, 8]. `sbox(round, x)` maps the round's 20 S-box inputs [batch, 20, 8]
    (0..15: the state, 16..19: RotWord of the last key word) to their outputs; `constants`: add Rcon (the public
    component of a sharing carries the constants, the lambda components do not)."""
    state = pt_bits ^ key_bits  # AddRoundKey 0
    rk = key_bits
    for rnd in range(1, ROUNDS + 1):
        rot = rk[:, [13, 14, 15, 12]]  # RotWord of w[4 rnd - 1] = bytes 12..15 of the previous round key
        y = sbox(rnd - 1, np.concatenate([state, rot], axis=1))
        sub, tmp = y[:, :16], y[:, 16:].copy()
        if constants:
            tmp[:, 0] ^= bytes_to_bits(np.uint8(RCON[rnd - 1]))
        nk = np.empty_like(rk)
        nk[:, 0:4] = rk[:, 0:4] ^ tmp
        for w in range(1, 4):
            nk[:, 4 * w:4 * w + 4] = rk[:, 4 * w:4 * w + 4] ^ nk[:, 4 * w - 4:4 * w]
        rk = nk
        state = _shift_rows(sub)
        if rnd != ROUNDS:
            state = _mix_columns(state)
        state = state ^ rk
    return state


class Aes128Flute:
    """Two-party evaluation of `batch` AES-128 instances; `sbox_tables`: the 8 tables of sbox8.lut (row order of the
    reference, LutTemplate convention)."""

    def __init__(self, sbox_tables, batch):
        self.batch = batch
        self.template = LutTemplate([(tuple(range(8)), t) for t in sbox_tables])
        self.units = ROUNDS * batch * SBOXES_PER_ROUND  # unit u = (round * batch + instance) * 20 + sbox
        self.setup_circuit = self.template.setup_circuit(self.units)

    # ---- offline: input-independent. Every wire's lambda is fixed; S-box outputs get fresh random lambdas.
    def offline(self, rng):
        b = self.batch
        self.lam_pt = [rng.integers(0, 2, (b, 16, 8), dtype=np.uint8) for _ in range(2)]
        self.lam_key = [rng.integers(0, 2, (b, 16, 8), dtype=np.uint8) for _ in range(2)]
        self.lam_sbox_in = [np.zeros((ROUNDS, b, SBOXES_PER_ROUND, 8), dtype=np.uint8) for _ in range(2)]
        self.lam_sbox_out = [rng.integers(0, 2, (ROUNDS, b, SBOXES_PER_ROUND, 8), dtype=np.uint8) for _ in range(2)]
        self.lam_ct = []
        for p in range(2):
            def sbox(rnd, x, p=p):
                self.lam_sbox_in[p][rnd] = x
                return self.lam_sbox_out[p][rnd]
            self.lam_ct.append(aes128_walk(self.lam_pt[p], self.lam_key[p], sbox, constants=False))
        # setup-circuit inputs of unit u: lambda shares of addr[0..7] = of byte bits 7..0 (NOT leaves lambda alone)
        return [self.lam_sbox_in[p][..., ::-1].reshape(-1) for p in range(2)]

    # ---- online
    def online(self, and_circuits, pt_bytes, key_bytes):
        """and_circuits: the two parties' evaluated AndCircuit of self.setup_circuit. Returns the ciphertext bytes
        [batch, 16] reconstructed from the public Delta and both lambda shares of the output wires."""
        b = self.batch
        d_pt = bytes_to_bits(pt_bytes) ^ self.lam_pt[0] ^ self.lam_pt[1]
        d_key = bytes_to_bits(key_bytes) ^ self.lam_key[0] ^ self.lam_key[1]
        self.rounds_run = 0

        def sbox(rnd, dx):
            rows = (255 - bits_to_bytes(dx)).astype(np.uint8).reshape(b * SBOXES_PER_ROUND, 1).repeat(8, axis=1)
            first = rnd * b * SBOXES_PER_ROUND
            d = [online_delta_shares(and_circuits[p], self.template, b * SBOXES_PER_ROUND, rows,
                                     self.lam_sbox_out[p][rnd].reshape(-1, 8), p, first_instance=first) for p in range(2)]
            self.rounds_run += 1
            return (d[0] ^ d[1]).reshape(b, SBOXES_PER_ROUND, 8)  # Msg::Delta exchanged, public Delta of dout[0..7]

        d_ct = aes128_walk(d_pt, d_key, sbox, constants=True)
        return bits_to_bytes(d_ct ^ self.lam_ct[0] ^ self.lam_ct[1])
