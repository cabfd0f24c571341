//! Raw FFI bindings to `libflute_sot` (C ABI: `include/flute_sot.h` of the engine's repository), written by hand in the
//! style of `crates/bitpolymul-sys/src/bindings/*.rs`. One declaration per symbol of the header; see the header for the
//! reference call site each one replaces and INTEGRATION.md for the three places in `zappot` / `gmw` that call them.
//!
//! `Block` of zappot is `#[repr(C, align(16))] struct Block { data: u128 }` (zappot/src/util/block.rs:17-21): slices of
//! `Block`, `[Block; 2]` and `Array2<Block>::as_slice()` are passed as `*const c_void` / `*mut c_void` unchanged.
//! Every function returns 0 or a negative `SOT_E*` code; `sot_last_error()` gives the thread's message.
//! A context serialises its callers internally: the two directions of a `SilentMtProvider` may call into one `SotCtx`
//! from two threads (`tokio::join!`, gmw/src/mul_triple/silent_ot.rs:87-91).
#![allow(non_camel_case_types, non_snake_case)]
use std::os::raw::{c_char, c_int, c_void};

pub const SOT_OK: c_int = 0;
pub const SOT_EINVAL: c_int = -1;
pub const SOT_EDEVICE: c_int = -2;
pub const SOT_ENOMEM: c_int = -3;
pub const SOT_ESTATE: c_int = -4;
pub const SOT_ROLE_SENDER: c_int = 0;
pub const SOT_ROLE_RECEIVER: c_int = 1;
pub const SOT_OUT_B: c_int = 0;
pub const SOT_OUT_A: c_int = 1;
pub const SOT_OUT_SENDER_MSGS: c_int = 2;
pub const SOT_OUT_RECEIVER_MSGS: c_int = 3;

/// `sot_conf`: QuasiCyclicConf + PprfConfig (quasi_cyclic_encode.rs:156-233, pprf.rs:40-44)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SotConf {
    pub n: u64,
    pub scaler: u64,
    pub P: u64,
    pub parts: u64,
    pub size_per: u64,
    pub N2: u64,
    pub N: u64,
    pub depth: u64,
    pub n_blocks: u64,
    pub n64: u64,
    pub cols: u64,
    pub base_ot_count: u64,
}

/// `sot_shard_info`: what rank `rank` of a sharded expansion owns
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct SotShardInfo {
    pub rank: u32,
    pub world: u32,
    pub set0: u32,
    pub nsets_local: u32,
    pub ggm_replicated: u32,
    pub rows_per_rank: u32,
    pub k0: u64,
    pub n_local: u64,
    pub chunk: u64,
    pub x1_bytes_per_peer: u64,
    pub x2_bytes_per_peer: u64,
}

#[repr(C)]
pub struct SotCtx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct SotAndCirc {
    _private: [u8; 0],
}
#[repr(C)]
pub struct SotAlszCtx {
    _private: [u8; 0],
}

#[link(name = "flute_sot")]
extern "C" {
    pub fn sot_last_error() -> *const c_char;
    pub fn sot_device_count() -> c_int;
    pub fn sot_configure(n: u64, scaler: u64, sec_param: u64, out: *mut SotConf) -> c_int;
    pub fn sot_ctx_create(n: u64, scaler: u64, device: c_int, out: *mut *mut SotCtx) -> c_int;
    pub fn sot_ctx_destroy(ctx: *mut SotCtx);
    pub fn sot_ctx_conf(ctx: *const SotCtx, out: *mut SotConf) -> c_int;
    pub fn sot_ctx_device_bytes(ctx: *const SotCtx) -> u64;
    pub fn sot_ctx_sync(ctx: *mut SotCtx) -> c_int;
    pub fn sot_ctx_stream(ctx: *mut SotCtx) -> *mut c_void;
    pub fn sot_ctx_set_stream(ctx: *mut SotCtx, hip_stream: *mut c_void) -> c_int;
    pub fn sot_ctx_set_async(ctx: *mut SotCtx, on: c_int) -> c_int;
    pub fn sot_treegrp_sums_blocks(c: *const SotConf) -> usize;
    pub fn sot_treegrp_last_blocks(c: *const SotConf) -> usize;
    pub fn sot_sender_expand(ctx: *mut SotCtx, base_ots: *const c_void, delta: *const c_void, seed: *const c_void, num_threads: u64, tg_sums_out: *mut c_void, tg_last_out: *mut c_void) -> c_int;
    pub fn sot_receiver_expand(ctx: *mut SotCtx, base_ots: *const c_void, choices: *const u8, tg_sums: *const c_void, tg_last: *const c_void) -> c_int;
    pub fn sot_receiver_points(c: *const SotConf, choices: *const u8, points_out: *mut u64) -> c_int;
    pub fn sot_download_rT(ctx: *mut SotCtx, role: c_int, rT_out: *mut c_void) -> c_int;
    pub fn sot_qc_encode(ctx: *mut SotCtx, role: c_int, points: *const u64, npoints: u64) -> c_int;
    pub fn sot_qc_encode2(ctx: *mut SotCtx, points: *const u64, npoints: u64, packing: c_int, C_out: *mut u8) -> c_int;
    pub fn sot_download_encoded(ctx: *mut SotCtx, role: c_int, blocks_out: *mut c_void) -> c_int;
    pub fn sot_sender_hash(ctx: *mut SotCtx, delta: *const c_void, msgs_out: *mut c_void) -> c_int;
    pub fn sot_receiver_hash(ctx: *mut SotCtx, msgs_out: *mut c_void, choice_words_out: *mut u64) -> c_int;
    pub fn sot_mt_bits(ctx: *mut SotCtx, delta: *const c_void, a_out: *mut u64, b_out: *mut u64, c_out: *mut u64) -> c_int;
    pub fn sot_mt_role_bits_to_device(ctx: *mut SotCtx, role: c_int, delta: *const c_void, dst_device: *mut c_void, dst_bytes: u64) -> c_int;
    pub fn sot_mt_bits_to_device(ctx: *mut SotCtx, dst_device: *mut c_void, dst_bytes: u64) -> c_int;
    pub fn sot_blocks_to_device(ctx: *mut SotCtx, what: c_int, delta: *const c_void, dst_device: *mut c_void, dst_bytes: u64) -> c_int;
    pub fn sot_ctx_create_sharded(n: u64, scaler: u64, device: c_int, rank: u32, world: u32, ggm_replicated: c_int, out: *mut *mut SotCtx) -> c_int;
    pub fn sot_shard_info_get(ctx: *const SotCtx, out: *mut SotShardInfo) -> c_int;
    pub fn sot_shard_pack_leaves(ctx: *mut SotCtx, send_dev: *mut c_void) -> c_int;
    pub fn sot_shard_pack_leaves_role(ctx: *mut SotCtx, send_dev: *mut c_void, roles: u32) -> c_int;
    pub fn sot_shard_unpack_leaves(ctx: *mut SotCtx, recv_dev: *const c_void) -> c_int;
    pub fn sot_shard_local_leaves(ctx: *mut SotCtx) -> c_int;
    pub fn sot_shard_qc_encode(ctx: *mut SotCtx, points: *const u64, npoints: u64, send_dev: *mut c_void) -> c_int;
    pub fn sot_shard_unpack_encoded(ctx: *mut SotCtx, recv_dev: *const c_void) -> c_int;
    pub fn sot_and_circ_create(ctx: *mut SotCtx, n_inputs: u32, n_gates: u32, n_layers: u32, layer_counts: *const u32, x_src: *const u32, y_src: *const u32, out: *mut *mut SotAndCirc) -> c_int;
    pub fn sot_and_circ_destroy(circ: *mut SotAndCirc);
    pub fn sot_and_circ_set_inputs(circ: *mut SotAndCirc, input_words: *const u64) -> c_int;
    pub fn sot_and_circ_layer_msg(circ: *mut SotAndCirc, layer: u32, d_out: *mut u8, e_out: *mut u8) -> c_int;
    pub fn sot_and_circ_layer_eval(circ: *mut SotAndCirc, layer: u32, party_id: u32, d_peer: *const u8, e_peer: *const u8) -> c_int;
    pub fn sot_and_circ_layer_eval_peer(circ: *mut SotAndCirc, layer: u32, party_id: u32, peer: *mut SotAndCirc) -> c_int;
    pub fn sot_and_circ_layer_eval_msg(circ: *mut SotAndCirc, layer: u32, party_id: u32, d_peer: *const u8, e_peer: *const u8, d_next_out: *mut u8, e_next_out: *mut u8) -> c_int;
    pub fn sot_and_circ_run_layers_peer(circ0: *mut SotAndCirc, circ1: *mut SotAndCirc, first: u32, last: u32) -> c_int;
    pub fn sot_and_circ_values(circ: *mut SotAndCirc, bits_out: *mut u8) -> c_int;
    pub fn sot_lut_online_eval(circ: *mut SotAndCirc, first_instance: u32, batch: u32, n_luts: u32, n_in: u32, arity: *const u32, tt: *const u32, lsrc: *const u32, tcount: *const u32, masked_in: *const u8, lam_out: *const u8, party_id: u32, delta_share_out: *mut u8) -> c_int;
    pub fn sot_lut_online_eval_dev(circ: *mut SotAndCirc, first_instance: u32, batch: u32, n_luts: u32, n_in: u32, arity: *const u32, tt: *const u32, lsrc: *const u32, tcount: *const u32, masked_in_dev: *const c_void, lam_out_dev: *const c_void, party_id: u32, delta_share_dev: *mut c_void) -> c_int;
    pub fn sot_mt_remaining(ctx: *const SotCtx) -> u64;
    pub fn sot_alsz_receive(device: c_int, base_ots: *const c_void, choice_bytes: *const u8, count: u64, u_rows_out: *mut u8, msgs_out: *mut c_void) -> c_int;
    pub fn sot_alsz_send(device: c_int, base_ots: *const c_void, base_choices: *const c_void, u_rows: *const u8, count: u64, msgs_out: *mut c_void) -> c_int;
    pub fn sot_alsz_ctx_create(device: c_int, max_count: u64, out: *mut *mut SotAlszCtx) -> c_int;
    pub fn sot_alsz_ctx_destroy(actx: *mut SotAlszCtx);
    pub fn sot_alsz_ctx_receive(actx: *mut SotAlszCtx, base_ots: *const c_void, choice_bytes: *const u8, choices_on_device: c_int, count: u64, u_rows_out: *mut u8, msgs_out: *mut c_void) -> c_int;
    pub fn sot_alsz_ctx_send(actx: *mut SotAlszCtx, base_ots: *const c_void, base_choices: *const c_void, u_rows: *const u8, u_dev: *const c_void, count: u64, msgs_out: *mut c_void) -> c_int;
    pub fn sot_alsz_ctx_msgs_dev(actx: *mut SotAlszCtx) -> *mut c_void;
    pub fn sot_alsz_ctx_u_dev(actx: *mut SotAlszCtx) -> *mut c_void;
    pub fn sot_alsz_ctx_sync(actx: *mut SotAlszCtx) -> c_int;
    pub fn sot_alsz_receive_noalloc(device: c_int, base_ots: *const c_void, choice_bytes: *const u8, count: u64, u_rows_out: *mut u8, msgs_out: *mut c_void) -> c_int;
    pub fn sot_alsz_noalloc_max_count() -> u64;
    pub fn sot_random_silent_send(ctx: *mut SotCtx, base_ots: *const c_void, delta: *const c_void, seed: *const c_void, num_threads: u64, tg_sums_out: *mut c_void, tg_last_out: *mut c_void, msgs_out: *mut c_void) -> c_int;
    pub fn sot_random_silent_receive(ctx: *mut SotCtx, base_ots: *const c_void, choices: *const u8, tg_sums: *const c_void, tg_last: *const c_void, msgs_out: *mut c_void, choice_words_out: *mut u64) -> c_int;
    pub fn sot_prof_enable(ctx: *mut SotCtx, on: c_int) -> c_int;
    pub fn sot_prof_reset(ctx: *mut SotCtx) -> c_int;
    pub fn sot_prof_kernels() -> c_int;
    pub fn sot_prof_name(kernel_id: c_int) -> *const c_char;
    pub fn sot_prof_read(ctx: *mut SotCtx, kernel_id: c_int, total_ms: *mut f64, launches: *mut u64) -> c_int;
    pub fn sot_transpose128(device: c_int, tiles_in: *const c_void, tiles_out: *mut c_void, ntiles: u64) -> c_int;
    pub fn sot_aes128_encrypt(device: c_int, key16: *const c_void, in_: *const c_void, out: *mut c_void, nblocks: u64) -> c_int;
    pub fn sot_cr_hash(device: c_int, in_: *const c_void, out: *mut c_void, nblocks: u64) -> c_int;
    pub fn sot_qc_dual_encode_host(ctx: *mut SotCtx, role: c_int, rT: *const c_void, points: *const u64, npoints: u64, blocks_out: *mut c_void) -> c_int;
    pub fn sot_qc_transform_row(ctx: *mut SotCtx, row_in: *const u32, stage: c_int, row_out: *mut u32) -> c_int;
    pub fn sot_qc_log2_capacity(ctx: *const SotCtx) -> u32;
}
