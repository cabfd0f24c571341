"""ctypes loader for libflute_sot.so (C ABI: include/flute_sot.h)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FLUTE_SOT_LIB: another build of the same library (profiling experiments, tools/exp.sh)
LIB_PATH = os.environ.get("FLUTE_SOT_LIB") or os.path.join(_HERE, "libflute_sot.so")
_LIB = None


class SotError(RuntimeError):
    pass


class Conf(C.Structure):
    """sot_conf == QuasiCyclicConf + PprfConfig (quasi_cyclic_encode.rs:156-233, pprf.rs:40-44)."""
    _fields_ = [(k, C.c_uint64) for k in
                ("n", "scaler", "P", "parts", "size_per", "N2", "N", "depth",
                 "n_blocks", "n64", "cols", "base_ot_count")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class ShardInfo(C.Structure):
    """sot_shard_info (include/flute_sot.h)"""
    _fields_ = [(k, C.c_uint32) for k in ("rank", "world", "set0", "nsets_local", "ggm_replicated", "rows_per_rank")] + \
               [(k, C.c_uint64) for k in ("k0", "n_local", "chunk", "x1_bytes_per_peer", "x2_bytes_per_peer")]


EXPORTS = [
    "sot_last_error", "sot_device_count", "sot_configure", "sot_ctx_create", "sot_ctx_destroy", "sot_ctx_conf",
    "sot_ctx_device_bytes", "sot_ctx_sync", "sot_ctx_stream", "sot_ctx_set_stream", "sot_ctx_set_async", "sot_treegrp_sums_blocks", "sot_treegrp_last_blocks",
    "sot_sender_expand", "sot_receiver_expand", "sot_receiver_points", "sot_download_rT", "sot_qc_encode",
    "sot_qc_encode2", "sot_download_encoded", "sot_sender_hash", "sot_receiver_hash", "sot_mt_bits", "sot_mt_bits_to_device", "sot_blocks_to_device", "sot_mt_role_bits_to_device", "sot_random_silent_send",
    "sot_random_silent_receive", "sot_prof_enable", "sot_prof_reset", "sot_prof_kernels", "sot_prof_name",
    "sot_prof_read", "sot_transpose128", "sot_aes128_encrypt", "sot_cr_hash", "sot_qc_dual_encode_host",
    "sot_qc_transform_row", "sot_qc_log2_capacity", "sot_and_circ_create", "sot_and_circ_destroy",
    "sot_and_circ_set_inputs", "sot_and_circ_layer_msg", "sot_and_circ_layer_eval", "sot_and_circ_layer_eval_peer",
    "sot_and_circ_layer_eval_msg", "sot_and_circ_run_layers_peer",
    "sot_and_circ_values", "sot_lut_online_eval", "sot_lut_online_eval_dev", "sot_mt_remaining", "sot_alsz_receive", "sot_alsz_send", "sot_alsz_receive_noalloc",
    "sot_alsz_noalloc_max_count", "sot_alsz_ctx_create", "sot_alsz_ctx_destroy", "sot_alsz_ctx_receive",
    "sot_alsz_ctx_send", "sot_alsz_ctx_msgs_dev", "sot_alsz_ctx_u_dev", "sot_alsz_ctx_sync", "sot_ctx_create_sharded", "sot_shard_info_get", "sot_shard_pack_leaves", "sot_shard_pack_leaves_role",
    "sot_shard_unpack_leaves", "sot_shard_local_leaves", "sot_shard_qc_encode", "sot_shard_unpack_encoded",
]


def load_library():
    """Loads the HIP library. Raises SotError if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise SotError("libflute_sot.so is missing: run `make -C flute_amd/csrc` (or __graft_entry__.build())")
    # PyTorch-ROCm ships its own libamdhip64, and the first HIP runtime a process loads serves every later user. If this
    # library pulled in /opt/rocm's first, a later `import torch` would find "No HIP GPUs" (seen when a test touched
    # torch only after the engine). So when torch is installed, let it load its runtime first; without torch the
    # library runs on the system runtime alone.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    P, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    L.sot_last_error.restype = C.c_char_p
    L.sot_device_count.restype = i32
    L.sot_configure.argtypes = [u64, u64, u64, C.POINTER(Conf)]
    L.sot_ctx_create.argtypes = [u64, u64, i32, C.POINTER(P)]
    L.sot_ctx_destroy.argtypes = [P]
    L.sot_ctx_destroy.restype = None
    L.sot_ctx_conf.argtypes = [P, C.POINTER(Conf)]
    L.sot_ctx_device_bytes.argtypes = [P]
    L.sot_ctx_device_bytes.restype = u64
    L.sot_ctx_sync.argtypes = [P]
    L.sot_ctx_stream.argtypes = [P]
    L.sot_ctx_stream.restype = P
    L.sot_ctx_set_stream.argtypes = [P, P]
    L.sot_ctx_set_async.argtypes = [P, i32]
    L.sot_treegrp_sums_blocks.argtypes = [C.POINTER(Conf)]
    L.sot_treegrp_sums_blocks.restype = C.c_size_t
    L.sot_treegrp_last_blocks.argtypes = [C.POINTER(Conf)]
    L.sot_treegrp_last_blocks.restype = C.c_size_t
    L.sot_sender_expand.argtypes = [P, P, P, P, u64, P, P]
    L.sot_receiver_expand.argtypes = [P, P, P, P, P]
    L.sot_receiver_points.argtypes = [C.POINTER(Conf), P, P]
    L.sot_download_rT.argtypes = [P, i32, P]
    L.sot_qc_encode.argtypes = [P, i32, P, u64]
    L.sot_qc_encode2.argtypes = [P, P, u64, i32, P]
    L.sot_download_encoded.argtypes = [P, i32, P]
    L.sot_sender_hash.argtypes = [P, P, P]
    L.sot_receiver_hash.argtypes = [P, P, P]
    L.sot_mt_bits.argtypes = [P, P, P, P, P]
    L.sot_mt_bits_to_device.argtypes = [P, P, u64]
    L.sot_blocks_to_device.argtypes = [P, i32, P, P, u64]
    L.sot_random_silent_send.argtypes = [P, P, P, P, u64, P, P, P]
    L.sot_random_silent_receive.argtypes = [P, P, P, P, P, P, P]
    L.sot_prof_enable.argtypes = [P, i32]
    L.sot_prof_reset.argtypes = [P]
    L.sot_prof_kernels.restype = i32
    L.sot_prof_name.argtypes = [i32]
    L.sot_prof_name.restype = C.c_char_p
    L.sot_prof_read.argtypes = [P, i32, C.POINTER(C.c_double), C.POINTER(u64)]
    L.sot_transpose128.argtypes = [i32, P, P, u64]
    L.sot_aes128_encrypt.argtypes = [i32, P, P, P, u64]
    L.sot_cr_hash.argtypes = [i32, P, P, u64]
    L.sot_qc_dual_encode_host.argtypes = [P, i32, P, P, u64, P]
    L.sot_qc_transform_row.argtypes = [P, P, i32, P]
    L.sot_qc_log2_capacity.argtypes = [P]
    L.sot_qc_log2_capacity.restype = C.c_uint32
    u32 = C.c_uint32
    L.sot_and_circ_create.argtypes = [P, u32, u32, u32, P, P, P, C.POINTER(P)]
    L.sot_and_circ_destroy.argtypes = [P]
    L.sot_and_circ_destroy.restype = None
    L.sot_and_circ_set_inputs.argtypes = [P, P]
    L.sot_and_circ_layer_msg.argtypes = [P, u32, P, P]
    L.sot_and_circ_layer_eval.argtypes = [P, u32, u32, P, P]
    L.sot_and_circ_layer_eval_peer.argtypes = [P, u32, u32, P]
    L.sot_and_circ_layer_eval_msg.argtypes = [P, u32, u32, P, P, P, P]
    L.sot_and_circ_run_layers_peer.argtypes = [P, P, u32, u32]
    L.sot_lut_online_eval_dev.argtypes = [P, u32, u32, u32, u32, P, P, P, P, P, P, u32, P]
    L.sot_and_circ_values.argtypes = [P, P]
    L.sot_lut_online_eval.argtypes = [P, u32, u32, u32, u32, P, P, P, P, P, P, u32, P]
    L.sot_mt_role_bits_to_device.argtypes = [P, i32, P, P, u64]
    L.sot_mt_remaining.argtypes = [P]
    L.sot_alsz_receive.argtypes = [i32, P, P, u64, P, P]
    L.sot_alsz_send.argtypes = [i32, P, P, P, u64, P]
    L.sot_alsz_receive_noalloc.argtypes = [i32, P, P, u64, P, P]
    L.sot_alsz_noalloc_max_count.restype = u64
    L.sot_alsz_ctx_create.argtypes = [i32, u64, P]
    L.sot_alsz_ctx_destroy.argtypes = [P]
    L.sot_alsz_ctx_destroy.restype = None
    L.sot_alsz_ctx_receive.argtypes = [P, P, P, i32, u64, P, P]
    L.sot_alsz_ctx_send.argtypes = [P, P, P, P, P, u64, P]
    L.sot_alsz_ctx_msgs_dev.argtypes = [P]
    L.sot_alsz_ctx_msgs_dev.restype = P
    L.sot_alsz_ctx_u_dev.argtypes = [P]
    L.sot_alsz_ctx_u_dev.restype = P
    L.sot_alsz_ctx_sync.argtypes = [P]
    L.sot_ctx_create_sharded.argtypes = [u64, u64, i32, C.c_uint32, C.c_uint32, i32, C.POINTER(P)]
    L.sot_shard_info_get.argtypes = [P, C.POINTER(ShardInfo)]
    L.sot_shard_pack_leaves.argtypes = [P, P]
    L.sot_shard_pack_leaves_role.argtypes = [P, P, C.c_uint32]
    L.sot_shard_unpack_leaves.argtypes = [P, P]
    L.sot_shard_local_leaves.argtypes = [P]
    L.sot_shard_qc_encode.argtypes = [P, P, u64, P]
    L.sot_shard_unpack_encoded.argtypes = [P, P]
    L.sot_mt_remaining.restype = u64
    for name in EXPORTS:
        fn = getattr(L, name)
        if fn.restype is C.c_int or name.startswith("sot_") and fn.restype == C.c_int:
            fn.restype = i32
    _LIB = L
    return L


def check(rc):
    if rc != 0:
        raise SotError("libflute_sot error %d: %s" % (rc, load_library().sot_last_error().decode()))


def device_count():
    return int(load_library().sot_device_count())
