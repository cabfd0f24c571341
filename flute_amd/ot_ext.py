"""Mirror of zappot::ot_ext::{Sender, Receiver} (crates/zappot/src/ot_ext/mod.rs:48-237, traits ExtROTSender /
ExtROTReceiver, crates/zappot/src/traits.rs:60-90) for the part after the 128 base OTs: PRG rows, 128 x n transpose,
correlation-robust hash -- on the GPU. The base OTs and the channel that carries the u rows (ExtOTMsg::URow) stay with
the host.

Two forms of each call: the one-shot C entry points (sot_alsz_send / sot_alsz_receive: a device context created and
destroyed around the call), used when no `max_count` is given, and a device-resident context (sot_alsz_ctx_*) that keeps
its buffers across calls and can leave the results in HBM (`to_host=False`)."""
import ctypes as C

import numpy as np

from .lib import check, load_library

BASE_OT_COUNT = 128


class _Ctx:
    """sot_alsz_ctx: device buffers for up to max_count extensions per call"""

    def __init__(self, device, max_count):
        self.L = load_library()
        self.h = C.c_void_p()
        check(self.L.sot_alsz_ctx_create(device, max_count, C.byref(self.h)))
        self.max_count = max_count

    def close(self):
        if self.h:
            self.L.sot_alsz_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.L.sot_alsz_ctx_sync(self.h))

    @property
    def msgs_dev(self):
        return self.L.sot_alsz_ctx_msgs_dev(self.h)

    @property
    def u_dev(self):
        return self.L.sot_alsz_ctx_u_dev(self.h)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Receiver:
    """ot_ext::Receiver: holds the base-OT SENDER's 128 message pairs"""

    def __init__(self, base_ot_pairs, device=0, max_count=None):
        self.base = np.ascontiguousarray(base_ot_pairs, dtype=np.uint64).reshape(BASE_OT_COUNT, 2, 2)
        self.device = device
        self.ctx = _Ctx(device, max_count) if max_count else None

    def receive_random(self, choice_bytes, count, to_host=True, choices_dev=None):
        """-> (u_rows [128, count/8] uint8 for the peer, msgs [count, 2] uint64); with to_host=False (context form only)
        nothing is copied back and (None, None) is returned: see ctx.u_dev / ctx.msgs_dev / ctx.sync().
        choices_dev: device address of the count/8 choice bytes instead of `choice_bytes`."""
        if count % 8:
            raise ValueError("Number of OT extensions must be multiple of 8")
        if choices_dev is None:
            ch = np.ascontiguousarray(np.frombuffer(bytes(choice_bytes), dtype=np.uint8))
            if len(ch) * 8 < count:
                raise ValueError("fewer choice bits than OT extensions")
        L = load_library()
        if self.ctx is None:
            if not to_host or choices_dev is not None:
                raise ValueError("device-resident results need a context: pass max_count")
            u = np.zeros((BASE_OT_COUNT, count // 8), dtype=np.uint8)
            out = np.zeros((count, 2), dtype=np.uint64)
            check(L.sot_alsz_receive(self.device, _ptr(self.base), _ptr(ch), count, _ptr(u), _ptr(out)))
            return u, out
        u = np.zeros((BASE_OT_COUNT, count // 8), dtype=np.uint8) if to_host else None
        out = np.zeros((count, 2), dtype=np.uint64) if to_host else None
        check(L.sot_alsz_ctx_receive(self.ctx.h, _ptr(self.base), C.c_void_p(choices_dev) if choices_dev is not None else _ptr(ch),
                                     1 if choices_dev is not None else 0, count, _ptr(u) if to_host else None,
                                     _ptr(out) if to_host else None))
        return u, out


class Sender:
    """ot_ext::Sender: holds the base-OT RECEIVER's 128 outputs and its choices (= delta)"""

    def __init__(self, base_ot_msgs, base_choices16, device=0, max_count=None):
        self.base = np.ascontiguousarray(base_ot_msgs, dtype=np.uint64).reshape(BASE_OT_COUNT, 2)
        self.choices = np.ascontiguousarray(np.frombuffer(bytes(base_choices16), dtype=np.uint8))
        assert self.choices.size == 16
        self.device = device
        self.ctx = _Ctx(device, max_count) if max_count else None

    def send_random(self, u_rows, count, to_host=True, u_dev=None):
        """-> msgs [count, 2, 2] uint64 (None with to_host=False). u_dev: device address of the peer's u-row matrix
        ([128][ceil(count/128)] Blocks, Receiver.ctx.u_dev on the same GPU) instead of `u_rows`."""
        if count % 8:
            raise ValueError("Number of OT extensions must be multiple of 8")
        L = load_library()
        if self.ctx is None:
            if not to_host or u_dev is not None:
                raise ValueError("device-resident operands need a context: pass max_count")
            u = np.ascontiguousarray(u_rows, dtype=np.uint8).reshape(BASE_OT_COUNT, count // 8)
            out = np.zeros((count, 2, 2), dtype=np.uint64)
            check(L.sot_alsz_send(self.device, _ptr(self.base), _ptr(self.choices), _ptr(u), count, _ptr(out)))
            return out
        u = None if u_dev is not None else np.ascontiguousarray(u_rows, dtype=np.uint8).reshape(BASE_OT_COUNT, count // 8)
        out = np.zeros((count, 2, 2), dtype=np.uint64) if to_host else None
        check(L.sot_alsz_ctx_send(self.ctx.h, _ptr(self.base), _ptr(self.choices), _ptr(u) if u is not None else None,
                                  C.c_void_p(u_dev) if u_dev is not None else None, count, _ptr(out) if to_host else None))
        return out
