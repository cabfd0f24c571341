"""Wire form of the one message the silent-OT hot path exchanges: pprf::Msg::TreeGrp (pprf.rs:46-49, 752-757), as the
reference's channel codec writes it. mpc_channel runs remoc with `codec-bincode` (crates/zappot/Cargo.toml:43), and
remoc's Bincode codec serialises with `bincode::DefaultOptions::new()` (libs/remoc/remoc/src/codec/bincode.rs:21,30):
little endian, VARINT integer encoding, no length limit. Under serde's derive that gives, per message:

    u32  variant index of Msg::TreeGrp = 0                           varint -> one byte 0x00
    u64  g (usize; first tree of the group, pprf.rs:128-131)         varint
    for keep in 0, 1:    sums[keep]: Vec<[Block; 8]>                 varint length (depth - 1 after :253-254), then per
                                                                     level 8 Blocks (fixed arrays carry no length)
    last_ots: Vec<[Block; 4]>                                        varint length (min(8, pnt_count - g), :223), 4 Blocks

and a Block is `struct Block { data: u128 }` (util/block.rs:17-21): one u128, which bincode's varint scheme writes as
    v < 251 -> [v];  v < 2^16 -> 0xfb, u16 LE;  v < 2^32 -> 0xfc, u32 LE;  v < 2^64 -> 0xfd, u64 LE;  else 0xfe, u128 LE
so a random Block takes 17 bytes, not 16, and an all-zero one takes 1.

Only the payload codec lives here. remoc's framing, multiplexing and the TCP/TLS transport are the reference's control
plane (out of scope); a Rust shim that links libflute_sot.so keeps its own `#[derive(Serialize)]` and never
needs this module. It exists for a Python host that is handed raw remoc payload bytes.

Parity status: "parity unpinned" -- the reference holds no byte fixture of a serialised TreeGrp and no Rust toolchain is
available to produce one; the format above is bincode 1.3.3's documented varint encoding, and tests/test_wire.py pins
the codec to hand-computed vectors of that rule plus round trips.
"""
import numpy as np

_TAG_U16, _TAG_U32, _TAG_U64, _TAG_U128 = 0xFB, 0xFC, 0xFD, 0xFE


def varint_u64(v):
    """bincode VarintEncoding of one unsigned integer < 2^64"""
    v = int(v)
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return bytes([_TAG_U16]) + v.to_bytes(2, "little")
    if v < 1 << 32:
        return bytes([_TAG_U32]) + v.to_bytes(4, "little")
    return bytes([_TAG_U64]) + v.to_bytes(8, "little")


def _read_varint(buf, pos):
    if pos >= len(buf):
        raise ValueError("bincode varint: truncated at offset %d" % pos)
    t = int(buf[pos])
    if t < 251:
        return t, pos + 1
    size = {_TAG_U16: 2, _TAG_U32: 4, _TAG_U64: 8, _TAG_U128: 16}.get(t)
    if size is None:
        raise ValueError("bincode varint: invalid tag byte 0x%02x at offset %d" % (t, pos))
    if pos + 1 + size > len(buf):
        raise ValueError("bincode varint: truncated at offset %d" % pos)
    return int.from_bytes(bytes(buf[pos + 1:pos + 1 + size]), "little"), pos + 1 + size


def encode_blocks(blocks):
    """blocks: (..., 2) uint64 (lo, hi) -> bytes, every Block as a varint u128, in C order"""
    b = np.ascontiguousarray(blocks, dtype=np.uint64).reshape(-1, 2)
    lo, hi = b[:, 0], b[:, 1]
    size = np.where(hi != 0, 16, np.where(lo >= 1 << 32, 8, np.where(lo >= 1 << 16, 4, np.where(lo >= 251, 2, 0))))
    tag = np.where(hi != 0, _TAG_U128, np.where(lo >= 1 << 32, _TAG_U64,
                                                np.where(lo >= 1 << 16, _TAG_U32, np.where(lo >= 251, _TAG_U16, lo))))
    off = np.concatenate([[0], np.cumsum(size + 1)])
    out = np.zeros(int(off[-1]), dtype=np.uint8)
    out[off[:-1]] = tag.astype(np.uint8)
    raw = b.view(np.uint8).reshape(-1, 16)  # little-endian bytes of the u128
    for s in (2, 4, 8, 16):
        idx = np.nonzero(size == s)[0]
        if len(idx):
            out[(off[idx] + 1)[:, None] + np.arange(s)[None, :]] = raw[idx, :s]
    return out.tobytes()


def decode_blocks(buf, pos, count):
    """count varint u128 values from buf[pos:] -> ((count, 2) uint64, new pos)"""
    buf = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    count, pos = int(count), int(pos)
    out = np.zeros((count, 2), dtype=np.uint64)
    end = pos + 17 * count
    if end <= len(buf) and np.all(buf[pos:end:17] == _TAG_U128):  # the common case: every Block >= 2^64
        raw = buf[pos:end].reshape(count, 17)[:, 1:]
        out[:] = np.ascontiguousarray(raw).view(np.uint64).reshape(count, 2)
        return out, end
    for i in range(count):
        v, pos = _read_varint(buf, pos)
        out[i, 0], out[i, 1] = v & 0xFFFFFFFFFFFFFFFF, v >> 64
    return out, pos


def encode_tree_grp(g, sums, last_ots):
    """One Msg::TreeGrp. g: first tree of the group; sums: (2, depth-1, 8, 2) uint64; last_ots: (k, 4, 2) uint64 with
    k = min(8, pnt_count - g) -> bytes"""
    sums = np.asarray(sums, dtype=np.uint64)
    last_ots = np.asarray(last_ots, dtype=np.uint64)
    if sums.ndim != 4 or sums.shape[0] != 2 or sums.shape[2:] != (8, 2):
        raise ValueError("sums must have shape (2, depth-1, 8, 2)")
    if last_ots.ndim != 3 or last_ots.shape[1:] != (4, 2) or last_ots.shape[0] > 8:
        raise ValueError("last_ots must have shape (<=8, 4, 2)")
    parts = [b"\x00", varint_u64(g)]
    for keep in range(2):
        parts += [varint_u64(sums.shape[1]), encode_blocks(sums[keep])]
    parts += [varint_u64(last_ots.shape[0]), encode_blocks(last_ots)]
    return b"".join(parts)


def decode_tree_grp(msg):
    """bytes of one Msg::TreeGrp -> (g, sums (2, L, 8, 2), last_ots (k, 4, 2)); raises ValueError on malformed input"""
    buf = np.frombuffer(bytes(msg), dtype=np.uint8)
    variant, pos = _read_varint(buf, 0)
    if variant != 0:
        raise ValueError("not a pprf::Msg::TreeGrp (variant index %d)" % variant)
    g, pos = _read_varint(buf, pos)
    sums = []
    for _ in range(2):
        n, pos = _read_varint(buf, pos)
        if n > len(buf):  # a length no well-formed message of this size can hold
            raise ValueError("TreeGrp: implausible sums length %d" % n)
        blk, pos = decode_blocks(buf, pos, n * 8)
        sums.append(blk.reshape(n, 8, 2))
    if sums[0].shape != sums[1].shape:
        raise ValueError("TreeGrp: sums[0] and sums[1] differ in length")
    k, pos = _read_varint(buf, pos)
    if k > 8:
        raise ValueError("TreeGrp: last_ots longer than a tree group")
    last, pos = decode_blocks(buf, pos, k * 4)
    if pos != len(buf):
        raise ValueError("TreeGrp: %d trailing bytes" % (len(buf) - pos))
    return g, np.stack(sums), last.reshape(k, 4, 2)


def _parts_depth(conf):  # a QuasiCyclicConf (silent_ot.py) or a dict with the same keys
    get = conf.get if isinstance(conf, dict) else lambda k: getattr(conf, k)
    return int(get("parts")), int(get("depth"))


def tree_grps_to_msgs(conf, tg_sums, tg_last):
    """The flat TreeGrp arrays of Engine.sender_expand (include/flute_sot.h: tg_sums [grp][keep][d][tree] Blocks,
    tg_last [grp][tree][4] Blocks) -> list of Msg::TreeGrp payloads, group 0 first. (The reference's sender threads
    emit them in a timing-dependent order, pprf.rs:128,256-258; the receiver keys them by `g`, :330-341.)"""
    parts, depth = _parts_depth(conf)
    ngrp = (parts + 7) // 8
    s = np.asarray(tg_sums, dtype=np.uint64).reshape(ngrp, 2, depth - 1, 8, 2)
    l = np.asarray(tg_last, dtype=np.uint64).reshape(ngrp, 8, 4, 2)
    return [encode_tree_grp(8 * i, s[i], l[i, :min(8, parts - 8 * i)]) for i in range(ngrp)]


def msgs_to_tree_grps(conf, msgs):
    """Inverse of tree_grps_to_msgs for messages in ANY arrival order; every group must arrive exactly once."""
    parts, depth = _parts_depth(conf)
    ngrp = (parts + 7) // 8
    s = np.zeros((ngrp, 2, depth - 1, 8, 2), dtype=np.uint64)
    l = np.zeros((ngrp, 8, 4, 2), dtype=np.uint64)
    seen = set()
    for m in msgs:
        g, sums, last = decode_tree_grp(m)
        if g % 8 or g >= parts or g in seen:
            raise ValueError("TreeGrp: unexpected or repeated group %d" % g)
        if sums.shape[1] != depth - 1 or last.shape[0] != min(8, parts - g):
            raise ValueError("TreeGrp %d: lengths do not match the configuration" % g)
        seen.add(g)
        s[g // 8] = sums
        l[g // 8, :last.shape[0]] = last
    if len(seen) != ngrp:
        raise ValueError("TreeGrp: %d of %d groups received" % (len(seen), ngrp))
    return s, l
