"""flute_amd/wire.py: pprf::Msg::TreeGrp in remoc's bincode (DefaultOptions = varint) encoding.
"parity unpinned": no reference-produced bytes exist (no Rust toolchain, no fixture in the reference); the vectors below
are computed by hand from bincode 1.3.3's documented VarintEncoding rule."""
import numpy as np
import pytest

from flute_amd import wire


def blk(v):
    return np.array([v & 0xFFFFFFFFFFFFFFFF, v >> 64], dtype=np.uint64)


@pytest.mark.parametrize("v,hexs", [
    (0, "00"), (250, "fa"), (251, "fbfb00"), (0xFFFF, "fbffff"), (0x10000, "fc00000100"),
    (0xFFFFFFFF, "fcffffffff"), (0x100000000, "fd0000000001000000"), ((1 << 64) - 1, "fd" + "ff" * 8),
    (1 << 64, "fe" + "00" * 8 + "01" + "00" * 7), ((1 << 128) - 1, "fe" + "ff" * 16),
    (0x0102030405060708090A0B0C0D0E0F10, "fe100f0e0d0c0b0a090807060504030201"),
])
def test_block_varint_vectors(v, hexs):
    enc = wire.encode_blocks(blk(v))
    assert enc.hex() == hexs
    dec, pos = wire.decode_blocks(enc, 0, 1)
    assert pos == len(enc) and (dec[0] == blk(v)).all()


def test_usize_varint():
    assert wire.varint_u64(8).hex() == "08"
    assert wire.varint_u64(256).hex() == "fb0001"
    assert wire.varint_u64(1 << 40).hex() == "fd0000000000010000"


def test_tree_grp_hand_vector():
    # depth 2 (one level of sums), g = 8, two trees in the group's tail; small Blocks so the bytes stay readable
    sums = np.zeros((2, 1, 8, 2), dtype=np.uint64)
    sums[0, 0, :, 0] = np.arange(8)
    sums[1, 0, :, 0] = 300 + np.arange(8)
    last = np.zeros((2, 4, 2), dtype=np.uint64)
    last[1, 3] = blk(1 << 100)
    msg = wire.encode_tree_grp(8, sums, last)
    want = ("00" "08"  # variant 0, g = 8
            "01" "0001020304050607"  # sums[0]: 1 level, Blocks 0..7
            "01" + "".join("fb%02x01" % (0x2c + i) for i in range(8)) +  # sums[1]: 300.. = 0x012c.. as u16 LE
            "02" + "00" * 7 + "fe" + "00" * 12 + "10000000")  # last_ots: 2 entries; the last Block = 2^100
    assert msg.hex() == want
    g, s, l = wire.decode_tree_grp(msg)
    assert g == 8 and (s == sums).all() and (l == last).all()


def test_round_trip_random_and_shuffled():
    rng = np.random.default_rng(5)
    conf = {"parts": 24, "depth": 9}
    s = rng.integers(0, 2 ** 64, size=(3, 2, 8, 8, 2), dtype=np.uint64)
    l = rng.integers(0, 2 ** 64, size=(3, 8, 4, 2), dtype=np.uint64)
    s[1, 0, 3, 2] = 0  # a zero Block and a small one: the 17-byte fast path must not be assumed
    s[2, 1, 0, 0] = blk(70000)
    msgs = wire.tree_grps_to_msgs(conf, s, l)
    assert len(msgs[0]) == 2 + 2 * (1 + 64 * 17) + 1 + 32 * 17  # all Blocks >= 2^64: 17 bytes each
    assert len(msgs[1]) == len(msgs[0]) - 16 and len(msgs[2]) == len(msgs[0]) - 12
    for order in ([0, 1, 2], [2, 0, 1]):
        s2, l2 = wire.msgs_to_tree_grps(conf, [msgs[i] for i in order])
        assert (s2 == s).all() and (l2 == l).all()


def test_malformed_messages_are_rejected():
    conf = {"parts": 16, "depth": 3}
    s = np.ones((2, 2, 2, 8, 2), dtype=np.uint64)
    l = np.ones((2, 8, 4, 2), dtype=np.uint64)
    msgs = wire.tree_grps_to_msgs(conf, s, l)
    with pytest.raises(ValueError):
        wire.msgs_to_tree_grps(conf, msgs[:1])  # a group missing
    with pytest.raises(ValueError):
        wire.msgs_to_tree_grps(conf, [msgs[0], msgs[0]])  # repeated
    with pytest.raises(ValueError):
        wire.decode_tree_grp(msgs[0][:-1])  # truncated
    with pytest.raises(ValueError):
        wire.decode_tree_grp(msgs[0] + b"\x00")  # trailing byte
    with pytest.raises(ValueError):
        wire.decode_tree_grp(b"\x01" + msgs[0][1:])  # another enum variant
    with pytest.raises(ValueError):
        wire.decode_tree_grp(b"\x00\x00\xff")  # 0xff is no varint tag
    with pytest.raises(ValueError):
        wire.msgs_to_tree_grps({"parts": 16, "depth": 4}, msgs)  # wrong depth for this configuration
