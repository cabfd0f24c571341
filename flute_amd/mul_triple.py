"""Mirror of gmw::mul_triple::{MulTriples, silent_ot::SilentMtProvider}
(crates/gmw/src/mul_triple/mod.rs:17-35,172-185 and silent_ot.rs:15-147)."""
import numpy as np

from .silent_ot import Engine, Receiver, Sender


class MulTriples:
    """Three bit vectors (BitVec<usize, Lsb0>) a, b, c of equal length -- mod.rs:30-35."""

    def __init__(self, a, b, c, n):
        self.a, self.b, self.c, self.n = a, b, c, n

    def __len__(self):
        return self.n

    @staticmethod
    def _bits(words, n):
        return np.unpackbits(words.view(np.uint8), bitorder="little")[:n]

    @staticmethod
    def _pack(bits):
        pad = (-len(bits)) % 64
        return np.packbits(np.concatenate([bits, np.zeros(pad, dtype=np.uint8)]), bitorder="little").view(np.uint64)

    @classmethod
    def from_gathered(cls, words, n):
        """The party's MT store out of dist.all_gather_words: words[rank] = a | b | c words of that rank's instance
        (n triples each); the instances are concatenated in rank order, bit-exactly (n need not be a multiple of 64)."""
        words = np.ascontiguousarray(np.asarray(words)).view(np.uint64)
        world, nw = words.shape[0], words.shape[1] // 3
        parts = [[cls._bits(words[r, i * nw:(i + 1) * nw], n) for r in range(world)] for i in range(3)]
        a, b, c = (cls._pack(np.concatenate(p)) for p in parts)
        return cls(a, b, c, world * n)

    def split_off_last(self, k):
        """mod.rs:172-185: consumers take triples from the END."""
        if k > self.n:
            raise ValueError("not enough multiplication triples")
        keep = self.n - k
        outs = []
        for name in ("a", "b", "c"):
            bits = self._bits(getattr(self, name), self.n)
            outs.append(self._pack(bits[keep:]))
            setattr(self, name, self._pack(bits[:keep]))
        self.n = keep
        return MulTriples(outs[0], outs[1], outs[2], k)


class SilentMtProvider:
    """gmw/src/mul_triple/silent_ot.rs. The base-OT phase (new_with_mult_type, :38-77) is out of scope:
    the provider is built from finished base OTs; the peer's TreeGrp messages are passed in explicitly."""

    def __init__(self, engine, sender_base_ots, receiver_base_ots, receiver_base_choices, num_threads):
        self.engine = engine
        self.configured_ots = engine.conf.n
        self.silent_sender = Sender.new_with_silent_base_ots(engine, sender_base_ots, num_threads)
        self.silent_receiver = Receiver.new_with_silent_base_ots(engine, receiver_base_ots, receiver_base_choices,
                                                                 num_threads)
        self.stored_mts = None
        self.tree_groups_out = None

    def send_phase(self, delta, seed):
        """First half of precompute_mts (:87): expand + encode the sender instance; returns the TreeGrp payload
        that the reference pushes through mpc_channel."""
        self.delta = np.asarray(delta, dtype=np.uint64)
        e = self.engine
        self.tree_groups_out = e.sender_expand(self.silent_sender.base_ots, delta, seed, self.silent_sender.num_threads)
        e.qc_encode(0)
        return self.tree_groups_out

    def receive_phase(self, peer_tree_groups):
        """Second half (:89-113): receiver instance with the peer's TreeGrps, then the fused hash + MT-bit kernel."""
        e = self.engine
        e.receiver_expand(self.silent_receiver.base_ots, self.silent_receiver.choices, *peer_tree_groups)
        e.qc_encode(1, self.silent_receiver.S)
        a, b, c = e.mt_bits(self.delta)
        self.stored_mts = MulTriples(a, b, c, self.configured_ots)

    def mts_available(self):
        return len(self.stored_mts) if self.stored_mts is not None else 0

    def request_mts(self, amount):
        """MTProvider::request_mts (:140-146)"""
        if self.stored_mts is None:
            raise RuntimeError("precompute_mts has not run (call send_phase / receive_phase)")
        return self.stored_mts.split_off_last(amount)
