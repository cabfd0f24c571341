"""TEST INFRASTRUCTURE ONLY: CPU restatement of the local work of the ALSZ13 OT extension,
crates/zappot/src/ot_ext/mod.rs (Sender::send_random :48-142, Receiver::receive_random :156-237), on top of the C
oracle's primitives (AesRng util/aes_rng.rs:101-136, transpose util/transpose.rs:5-14, cr_hash util/aes_hash.rs:26-30).
The base OTs (CO15) and the channel are out of scope: seeds, choices and the u rows are explicit arguments.
Pinned by the reference's own test `ot_ext` (:280-307: recv[i] == sent[i][0] for all-zero choices) restated in
tests/test_alsz.py, generalised to random choices; no reference-produced vector exists ("parity unpinned")."""
import numpy as np

from . import sot_oracle as orc

BASE_OT_COUNT = 128


def _prg_row(seed_block, nbytes):
    """AesRng::from_seed(seed).fill_bytes(row) (:101-106, :204-206)"""
    nblk = (nbytes + 15) // 16
    return orc.aes_rng_blocks(seed_block, 0, nblk).view(np.uint8).reshape(-1)[:nbytes].copy()


def receive_random(base_ots, choice_bytes, count):
    """base_ots: [128][2] Blocks as uint64 [128, 2, 2]; choice_bytes: count/8 bytes (Lsb0). Returns (u_rows [128, count/8]
    uint8, msgs [count, 2] uint64)."""
    assert count % 8 == 0
    cols = count // 8
    choices = np.frombuffer(bytes(choice_bytes), dtype=np.uint8)[:cols]
    t_mat = np.zeros((BASE_OT_COUNT, cols), dtype=np.uint8)
    u_rows = np.zeros((BASE_OT_COUNT, cols), dtype=np.uint8)
    for idx in range(BASE_OT_COUNT):
        s0, s1 = base_ots[idx]
        t_mat[idx] = _prg_row(s0, cols)  # prg0.fill_bytes(t_row)
        u_rows[idx] = _prg_row(s1, cols) ^ t_mat[idx] ^ choices  # *val ^= *rand_val ^ choice (:212-217)
    tt = np.frombuffer(orc.transpose_naive(t_mat.tobytes(), BASE_OT_COUNT, count), dtype=np.uint8)
    blocks = tt.reshape(count, 16).copy().view(np.uint64).reshape(count, 2)
    return u_rows, orc.cr_hash(blocks)


def send_random(base_ots, base_choices16, u_rows, count):
    """base_ots: [128] Blocks [128, 2] uint64 (the seeds the base-OT receiver got); base_choices16: the 128 base choice
    bits as 16 Lsb0 bytes (delta = Block of them, :88-90). Returns msgs [count, 2, 2] uint64."""
    assert count % 8 == 0
    cols = count // 8
    bc = np.frombuffer(bytes(base_choices16), dtype=np.uint8)
    delta = bc.copy().view(np.uint64).reshape(1, 2)
    v_mat = np.zeros((BASE_OT_COUNT, cols), dtype=np.uint8)
    for idx in range(BASE_OT_COUNT):
        v_mat[idx] = _prg_row(base_ots[idx], cols)
        r = (bc[idx >> 3] >> (idx & 7)) & 1
        if r:  # computes r_j * u_j (:118-126)
            v_mat[idx] ^= np.asarray(u_rows[idx], dtype=np.uint8)
    vt = np.frombuffer(orc.transpose_naive(v_mat.tobytes(), BASE_OT_COUNT, count), dtype=np.uint8)
    blocks = vt.reshape(count, 16).copy().view(np.uint64).reshape(count, 2)
    out = np.zeros((count, 2, 2), dtype=np.uint64)
    out[:, 0] = orc.cr_hash(blocks)
    out[:, 1] = orc.cr_hash(blocks ^ delta)
    return out


def fake_base(rng):
    """base OTs as both sides see them: pairs for the extension RECEIVER (base-OT sender), the chosen seed and the
    choices for the extension SENDER (base-OT receiver)"""
    pairs = rng.integers(0, 2 ** 64, (BASE_OT_COUNT, 2, 2), dtype=np.uint64)
    bits = rng.integers(0, 2, BASE_OT_COUNT, dtype=np.uint8)
    chosen = pairs[np.arange(BASE_OT_COUNT), bits]
    return pairs, chosen.copy(), np.packbits(bits, bitorder="little")
