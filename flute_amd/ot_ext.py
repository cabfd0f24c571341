"""Mirror of zappot::ot_ext::{Sender, Receiver} (crates/zappot/src/ot_ext/mod.rs:48-237, traits ExtROTSender /
ExtROTReceiver, crates/zappot/src/traits.rs:60-90) for the part after the 128 base OTs: PRG rows, 128 x n transpose,
correlation-robust hash -- on the GPU through sot_alsz_send / sot_alsz_receive. The base OTs and the channel that
carries the u rows (ExtOTMsg::URow) stay with the host."""
import ctypes as C

import numpy as np

from .lib import check, load_library

BASE_OT_COUNT = 128


class Receiver:
    """ot_ext::Receiver: holds the base-OT SENDER's 128 message pairs"""

    def __init__(self, base_ot_pairs, device=0):
        self.base = np.ascontiguousarray(base_ot_pairs, dtype=np.uint64).reshape(BASE_OT_COUNT, 2, 2)
        self.device = device

    def receive_random(self, choice_bytes, count):
        """-> (u_rows [128, count/8] uint8 for the peer, msgs [count, 2] uint64)"""
        ch = np.ascontiguousarray(np.frombuffer(bytes(choice_bytes), dtype=np.uint8))
        if count % 8 or len(ch) * 8 < count:
            raise ValueError("Number of OT extensions must be multiple of 8")
        u = np.zeros((BASE_OT_COUNT, count // 8), dtype=np.uint8)
        out = np.zeros((count, 2), dtype=np.uint64)
        check(load_library().sot_alsz_receive(self.device, self.base.ctypes.data_as(C.c_void_p),
                                              ch.ctypes.data_as(C.c_void_p), count, u.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p)))
        return u, out


class Sender:
    """ot_ext::Sender: holds the base-OT RECEIVER's 128 outputs and its choices (= delta)"""

    def __init__(self, base_ot_msgs, base_choices16, device=0):
        self.base = np.ascontiguousarray(base_ot_msgs, dtype=np.uint64).reshape(BASE_OT_COUNT, 2)
        self.choices = np.ascontiguousarray(np.frombuffer(bytes(base_choices16), dtype=np.uint8))
        assert self.choices.size == 16
        self.device = device

    def send_random(self, u_rows, count):
        """-> msgs [count, 2, 2] uint64"""
        if count % 8:
            raise ValueError("Number of OT extensions must be multiple of 8")
        u = np.ascontiguousarray(u_rows, dtype=np.uint8).reshape(BASE_OT_COUNT, count // 8)
        out = np.zeros((count, 2, 2), dtype=np.uint64)
        check(load_library().sot_alsz_send(self.device, self.base.ctypes.data_as(C.c_void_p),
                                           self.choices.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), count,
                                           out.ctypes.data_as(C.c_void_p)))
        return out
