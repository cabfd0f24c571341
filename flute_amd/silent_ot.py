"""Host-side mirror of zappot::silent_ot (crates/zappot/src/silent_ot/mod.rs) over libflute_sot.

Names and argument meaning follow the reference: QuasiCyclicConf::configure, Sender /
Receiver::new_with_silent_base_ots, random_silent_send / random_silent_receive,
correlated_silent_send / correlated_silent_receive. The mpc_channel hop of the reference
(pprf::Msg::TreeGrp, pprf.rs:256-258,330-341) is replaced by returning / accepting the flat TreeGrp
arrays: the Rust shim keeps the channel.

Blocks are numpy uint64 arrays with trailing dimension 2 (little-endian u128).
"""
import ctypes as C

import numpy as np

from .lib import Conf, ShardInfo, check, load_library

SECURITY_PARAM = 128  # silent_ot/mod.rs
ROLE_SENDER, ROLE_RECEIVER = 0, 1


class MultType:
    """silent_ot::MultType::QuasiCyclic{scaler} (the only value bench.rs:85 uses)."""

    def __init__(self, scaler=2):
        self.scaler = scaler


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _blk(b):
    return np.ascontiguousarray(np.asarray(b, dtype=np.uint64).reshape(2))


class QuasiCyclicConf:
    """QuasiCyclicConf::configure(num_ots, scaler, sec_param) -- quasi_cyclic_encode.rs:177-193."""

    def __init__(self, c):
        self.c = c
        for k, v in c.as_dict().items():
            setattr(self, k, v)
        self.requested_num_ots = self.n
        self.num_partitions = self.parts

    @staticmethod
    def configure(num_ots, scaler=2, sec_param=SECURITY_PARAM):
        c = Conf()
        check(load_library().sot_configure(num_ots, scaler, sec_param, C.byref(c)))
        return QuasiCyclicConf(c)

    def treegrp_shapes(self):
        grp = self.parts // 8
        return (grp, 2, self.depth - 1, 8, 2), (grp, 8, 4, 2)


def get_active_path(bits):
    """pprf.rs:701-706"""
    d = len(bits)
    p = 0
    for i, b in enumerate(bits):
        p |= (1 ^ int(b)) << (d - i - 1)
    return p


def interleave_point(point, tree_idx, total_trees):
    """pprf.rs:716-741 (InterleavedTransposed)"""
    num_sets = total_trees // 8
    return (tree_idx // 8) * 128 + (tree_idx % 8 + 8 * (point % 16)) + (point // 16) * num_sets * 128


def sample_base_choice_bits(conf, rng):
    """Receiver::sample_base_choice_bits -> pprf::Receiver::sample_choice_bits (pprf.rs:579-620):
    rejection-sample each tree's path until its interleaved point is < N2. rng: numpy Generator."""
    ch = np.zeros((conf.parts, conf.depth), dtype=np.uint8)
    for i in range(conf.parts):
        assert interleave_point(0, i, conf.parts) < conf.N2
        while True:
            ch[i] = rng.integers(0, 2, size=conf.depth, dtype=np.uint8)
            if interleave_point(get_active_path(ch[i]), i, conf.parts) < conf.N2:
                break
    return ch


def fake_base(conf, rng):
    """pprf.rs:858-873 test helper: fabricated base OTs (sender pairs, receiver picks, choices)."""
    cnt = conf.base_ot_count
    msg2 = rng.integers(0, 2 ** 64, size=(cnt, 2, 2), dtype=np.uint64)
    choices = sample_base_choice_bits(conf, rng)
    msg = msg2[np.arange(cnt), choices.reshape(-1).astype(np.int64)].copy()
    return msg2, msg, choices


class Engine:
    """One device context = one sender + one receiver instance of n OTs (what SilentMtProvider holds,
    gmw/src/mul_triple/silent_ot.rs:15-23)."""

    def __init__(self, num_ots, mult_type=None, device=0):
        self.lib = load_library()
        scaler = (mult_type or MultType()).scaler
        h = C.c_void_p()
        check(self.lib.sot_ctx_create(num_ots, scaler, device, C.byref(h)))
        self.h = h
        self._finish_init(device)

    def _finish_init(self, device):
        c = Conf()
        check(self.lib.sot_ctx_conf(self.h, C.byref(c)))
        self.conf = QuasiCyclicConf(c)
        self.device = device
        self.n_out = self.conf.n  # outputs of the hash stage held by this context (a sharded one holds n_local)
        self._circuits = []  # AndCircuit objects evaluated on this context (lut_setup.py)

    def close(self):
        for circ in getattr(self, "_circuits", []):
            circ.close()
        self._circuits = []
        if getattr(self, "h", None):
            self.lib.sot_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_bytes(self):
        return int(self.lib.sot_ctx_device_bytes(self.h))

    def stream(self):
        return self.lib.sot_ctx_stream(self.h)

    def sync(self):
        check(self.lib.sot_ctx_sync(self.h))

    def set_stream(self, hip_stream):
        """queue this context's work on the caller's hipStream_t (an int handle, e.g. torch.cuda.current_stream().cuda_stream;
        None: back to the context's own stream)"""
        check(self.lib.sot_ctx_set_stream(self.h, C.c_void_p(hip_stream) if hip_stream else None))

    def set_async(self, on=True):
        """stage calls that hand nothing to the host no longer wait for the stream (sot_ctx_set_async)"""
        check(self.lib.sot_ctx_set_async(self.h, 1 if on else 0))

    # ---- validation hook of the GF(2)[x] product (sot_qc_transform_row)
    def qc_log2_capacity(self):
        return int(self.lib.sot_qc_log2_capacity(self.h))

    def qc_transform_row(self, row_words, stage):
        m = self.qc_log2_capacity()
        row = np.ascontiguousarray(row_words, dtype=np.uint32)
        assert row.size == 1 << (m - 6)
        out = np.zeros(1 << (m - 5), dtype=np.uint32)
        check(self.lib.sot_qc_transform_row(self.h, _p(row), stage, _p(out)))
        return out

    # ---- per-kernel HIP-event timing (sot_prof_*)
    def prof_enable(self, on=True):
        check(self.lib.sot_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        check(self.lib.sot_prof_reset(self.h))

    def prof_read(self):
        """{kernel name: (total device ms, launches)} for kernels that ran."""
        out = {}
        for i in range(self.lib.sot_prof_kernels()):
            ms, cnt = C.c_double(), C.c_uint64()
            check(self.lib.sot_prof_read(self.h, i, C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[self.lib.sot_prof_name(i).decode()] = (ms.value, int(cnt.value))
        return out

    # ---- stage-level calls (shapes checked here: the kernels trust them)
    def sender_expand(self, base_ots, delta, seed, num_threads):
        cf = self.conf
        base_ots = np.ascontiguousarray(base_ots, dtype=np.uint64)
        if base_ots.shape != (cf.base_ot_count, 2, 2):
            raise ValueError("Wrong number of silent base ots")  # mod.rs:171-175
        s_shape, l_shape = cf.treegrp_shapes()
        sums = np.zeros(s_shape, dtype=np.uint64)
        last = np.zeros(l_shape, dtype=np.uint64)
        d, s = _blk(delta), _blk(seed)
        check(self.lib.sot_sender_expand(self.h, _p(base_ots), _p(d), _p(s), int(num_threads), _p(sums), _p(last)))
        return sums, last

    def receiver_expand(self, base_ots, choices, tg_sums, tg_last):
        cf = self.conf
        base_ots = np.ascontiguousarray(base_ots, dtype=np.uint64)
        choices = np.ascontiguousarray(choices, dtype=np.uint8)
        if base_ots.shape != (cf.base_ot_count, 2) or choices.size != cf.base_ot_count:
            raise ValueError("Wrong number of silent base ots / choices")
        s_shape, l_shape = cf.treegrp_shapes()
        tg_sums = np.ascontiguousarray(tg_sums, dtype=np.uint64)
        tg_last = np.ascontiguousarray(tg_last, dtype=np.uint64)
        if tg_sums.shape != s_shape or tg_last.shape != l_shape:
            raise ValueError("TreeGrp arrays have the wrong shape")
        check(self.lib.sot_receiver_expand(self.h, _p(base_ots), _p(choices), _p(tg_sums), _p(tg_last)))

    def points(self, choices):
        choices = np.ascontiguousarray(choices, dtype=np.uint8)
        pts = np.zeros(self.conf.parts, dtype=np.uint64)
        check(self.lib.sot_receiver_points(C.byref(self.conf.c), _p(choices), _p(pts)))
        return pts

    def download_rT(self, role):
        out = np.zeros((128, self.conf.cols, 2), dtype=np.uint64)
        check(self.lib.sot_download_rT(self.h, role, _p(out)))
        return out

    def qc_encode(self, role, points=None):
        if role == ROLE_RECEIVER:
            pts = np.ascontiguousarray(points, dtype=np.uint64)
            check(self.lib.sot_qc_encode(self.h, role, _p(pts), pts.size))
        else:
            check(self.lib.sot_qc_encode(self.h, role, None, 0))

    def qc_encode2(self, points, packed=True):
        """QuasiCyclicEncoder::dual_encode2 with the ChoiceBitPacking as an argument (quasi_cyclic_encode.rs:54-124).
        Returns None (packed) or the choice vector C as n bytes 0/1 (unpacked)."""
        pts = np.ascontiguousarray(points, dtype=np.uint64)
        if packed:
            check(self.lib.sot_qc_encode2(self.h, _p(pts), pts.size, 1, None))
            return None
        Cv = np.zeros(self.conf.n, dtype=np.uint8)
        check(self.lib.sot_qc_encode2(self.h, _p(pts), pts.size, 0, _p(Cv)))
        return Cv

    def qc_dual_encode_host(self, role, rT, points=None):
        rT = np.ascontiguousarray(rT, dtype=np.uint64)
        if rT.shape != (128, self.conf.cols, 2):
            raise ValueError("rT must be 128 x cols Blocks")
        out = np.zeros((self.conf.n, 2), dtype=np.uint64)
        if role == ROLE_RECEIVER:
            pts = np.ascontiguousarray(points, dtype=np.uint64)
            check(self.lib.sot_qc_dual_encode_host(self.h, role, _p(rT), _p(pts), pts.size, _p(out)))
        else:
            check(self.lib.sot_qc_dual_encode_host(self.h, role, _p(rT), None, 0, _p(out)))
        return out

    def download_encoded(self, role):
        out = np.zeros((self.n_out, 2), dtype=np.uint64)
        check(self.lib.sot_download_encoded(self.h, role, _p(out)))
        return out

    def sender_hash(self, delta, download=True):
        d = _blk(delta)
        out = np.zeros((self.n_out, 2, 2), dtype=np.uint64) if download else None
        check(self.lib.sot_sender_hash(self.h, _p(d), _p(out) if download else None))
        return out

    def receiver_hash(self, download=True):
        n = self.n_out
        out = np.zeros((n, 2), dtype=np.uint64) if download else None
        ch = np.zeros((n + 63) // 64, dtype=np.uint64)
        check(self.lib.sot_receiver_hash(self.h, _p(out) if download else None, _p(ch)))
        return out, ch

    def mt_bits(self, delta, download=True):
        n = self.n_out
        d = _blk(delta)
        nw = (n + 63) // 64
        if not download:
            check(self.lib.sot_mt_bits(self.h, _p(d), None, None, None))
            return None
        a = np.zeros(nw, dtype=np.uint64)
        b = np.zeros(nw, dtype=np.uint64)
        c = np.zeros(nw, dtype=np.uint64)
        check(self.lib.sot_mt_bits(self.h, _p(d), _p(a), _p(b), _p(c)))
        return a, b, c


    def mt_role_bits_to_device(self, role, delta, device_ptr, nbytes):
        """the MT bits of one direction (b | v for ROLE_SENDER, a | u for ROLE_RECEIVER; 2 * ceil(n / 64) words) into a
        caller-owned device buffer: sot_mt_role_bits_to_device, role split over two GPUs"""
        d = _blk(delta) if delta is not None else None
        check(self.lib.sot_mt_role_bits_to_device(self.h, role, _p(d) if d is not None else None, C.c_void_p(device_ptr),
                                                  nbytes))

    def mt_words(self):
        """3 * ceil(n / 64): length in u64 of the a | b | c buffer of mt_bits_to_device."""
        return 3 * ((self.n_out + 63) // 64)

    OUT_B, OUT_A, OUT_SENDER_MSGS, OUT_RECEIVER_MSGS = 0, 1, 2, 3

    def blocks_to_device(self, what, device_ptr, nbytes, delta=None):
        """B / A / Sender::hash / Receiver::hash output of this context (this rank's n_out outputs when sharded) into a
        caller-owned device buffer: sot_blocks_to_device"""
        d = _blk(delta) if delta is not None else None
        check(self.lib.sot_blocks_to_device(self.h, what, _p(d) if d is not None else None, C.c_void_p(device_ptr), nbytes))

    def mt_bits_to_device(self, device_ptr, nbytes):
        """a | b | c words of the last mt_bits() into a caller-owned device buffer (a torch tensor's data_ptr())."""
        check(self.lib.sot_mt_bits_to_device(self.h, C.c_void_p(device_ptr), nbytes))


class ShardedEngine(Engine):
    """Rank `rank` of ONE expansion sharded over `world` GPUs (include/flute_sot.h, sot_shard_*):
    this rank's tree sets in the GGM stage, its 128 / world bit rows of both roles in the QC stage, its n_local outputs
    in the hash stage. The two all-to-alls between the stages are run by the caller on device buffers
    (flute_amd.dist.ShardedPipeline over torch.distributed)."""

    def __init__(self, num_ots, rank, world, mult_type=None, device=0, ggm_replicated=False):
        self.lib = load_library()
        scaler = (mult_type or MultType()).scaler
        h = C.c_void_p()
        check(self.lib.sot_ctx_create_sharded(num_ots, scaler, device, rank, world, 1 if ggm_replicated else 0,
                                              C.byref(h)))
        self.h = h
        self._finish_init(device)
        info = ShardInfo()
        check(self.lib.sot_shard_info_get(self.h, C.byref(info)))
        self.info = info
        self.rank, self.world = rank, world
        self.ggm_replicated = bool(info.ggm_replicated)
        self.n_out = int(info.n_local)
        self.k0, self.chunk = int(info.k0), int(info.chunk)
        self.x1_bytes, self.x2_bytes = int(info.x1_bytes_per_peer), int(info.x2_bytes_per_peer)
        self.groups = range(int(info.set0), int(info.set0) + int(info.nsets_local))  # TreeGrps this rank makes / needs

    def pack_leaves(self, send_ptr, roles=3):
        check(self.lib.sot_shard_pack_leaves_role(self.h, C.c_void_p(send_ptr), roles))

    def unpack_leaves(self, recv_ptr):
        check(self.lib.sot_shard_unpack_leaves(self.h, C.c_void_p(recv_ptr)))

    def local_leaves(self):
        check(self.lib.sot_shard_local_leaves(self.h))

    def shard_qc_encode(self, points, send_ptr):
        pts = np.ascontiguousarray(points, dtype=np.uint64)
        check(self.lib.sot_shard_qc_encode(self.h, _p(pts), pts.size, C.c_void_p(send_ptr)))

    def unpack_encoded(self, recv_ptr):
        check(self.lib.sot_shard_unpack_encoded(self.h, C.c_void_p(recv_ptr)))


class Sender:
    """zappot::silent_ot::Sender (mod.rs:49-56)."""

    def __init__(self, engine, silent_base_ots, num_threads):
        self.engine = engine
        self.enc = engine.conf
        self.base_ots = np.ascontiguousarray(silent_base_ots, dtype=np.uint64)
        if self.base_ots.shape[0] != self.enc.base_ot_count:
            raise ValueError("Wrong number of silent base ots")
        self.num_threads = num_threads
        self.tree_groups = None

    @classmethod
    def new_with_silent_base_ots(cls, engine, silent_base_ots, num_threads):
        """mod.rs:166-190"""
        return cls(engine, silent_base_ots, num_threads)

    def correlated_silent_send(self, delta, seed):
        """mod.rs:217-247. `seed` is the Block pprf::Sender::expand draws at pprf.rs:96.
        Returns B; self.tree_groups = (sums, last) is what goes to the peer."""
        self.tree_groups = self.engine.sender_expand(self.base_ots, delta, seed, self.num_threads)
        self.engine.qc_encode(ROLE_SENDER)
        return self.engine.download_encoded(ROLE_SENDER)

    def random_silent_send(self, delta, seed):
        """mod.rs:193-211 (delta = rng.gen() there). Returns Vec<[Block; 2]> as [n, 2, 2]."""
        self.tree_groups = self.engine.sender_expand(self.base_ots, delta, seed, self.num_threads)
        self.engine.qc_encode(ROLE_SENDER)
        return self.engine.sender_hash(delta)


class Receiver:
    """zappot::silent_ot::Receiver (mod.rs:59-69)."""

    def __init__(self, engine, silent_base_ots, silent_base_choices, num_threads):
        self.engine = engine
        self.enc = engine.conf
        self.base_ots = np.ascontiguousarray(silent_base_ots, dtype=np.uint64)
        self.choices = np.ascontiguousarray(silent_base_choices, dtype=np.uint8)
        if self.base_ots.shape[0] != self.enc.base_ot_count:
            raise ValueError("Wrong number of silent base ots")
        self.num_threads = num_threads
        self.S = engine.points(self.choices)  # mod.rs:368

    @classmethod
    def new_with_silent_base_ots(cls, engine, silent_base_ots, silent_base_choices, num_threads):
        """mod.rs:353-384"""
        return cls(engine, silent_base_ots, silent_base_choices, num_threads)

    def correlated_silent_receive(self, tree_groups, choice_bit_packing=True):
        """mod.rs:410-435. ChoiceBitPacking::True: returns A with the choice bit packed in bit 0;
        ChoiceBitPacking::False: returns (A, C) with C the choice vector, one byte per OT (Option<Vec<u8>>)."""
        self.engine.receiver_expand(self.base_ots, self.choices, *tree_groups)
        if choice_bit_packing:
            self.engine.qc_encode(ROLE_RECEIVER, self.S)
            return self.engine.download_encoded(ROLE_RECEIVER)
        Cv = self.engine.qc_encode2(self.S, packed=False)
        return self.engine.download_encoded(ROLE_RECEIVER), Cv

    def random_silent_receive(self, tree_groups):
        """mod.rs:391-404. Returns (Vec<Block> [n,2], choices as ceil(n/64) u64 words, Lsb0)."""
        self.engine.receiver_expand(self.base_ots, self.choices, *tree_groups)
        self.engine.qc_encode(ROLE_RECEIVER, self.S)
        return self.engine.receiver_hash()
