/*
 * flute_sot.h -- C ABI of libflute_sot: MI355X (gfx950) silent-OT correlation engine for FLUTE.
 *
 * Drop-in boundary for the local, data-parallel expansion that runs after the base-OT seeds are
 * exchanged (zappot silent_ot sender/receiver + gmw SilentMtProvider bit derivation). The reference
 * has one FFI seam on this path (crates/bitpolymul-sys, a bindgen-style `extern "C"` block with raw
 * pointers and lengths); this header sits one level higher so that whole stages stay on the GPU.
 * A Rust `-sys` crate binds these symbols exactly as bitpolymul-sys binds bitpolymul2
 * (see INTEGRATION.md for the extern block and the call sites that change).
 *
 * Conventions
 *  - Block = 16 bytes, little-endian u128 (zappot/src/util/block.rs:17-21).
 *  - every function returns 0 on success, a negative SOT_E* code otherwise; sot_last_error()
 *    returns a thread-local message. Nothing calls exit()/abort() (the reference's C layer does:
 *    bitpolymul2/encode.c:102, bitpolymul.c:77).
 *  - "host" pointers are ordinary caller-owned memory; results stay resident in HBM inside the
 *    context until a *_download / output pointer asks for them.
 *  - every entry point that takes a context holds that context's lock for the duration of the call: the sender and the
 *    receiver direction of a party may be driven from two host threads at once, as gmw/src/mul_triple/silent_ot.rs:87-91
 *    does -- the calls serialise (both directions share the context's stream and scratch buffers); two contexts run
 *    concurrently. sot_and_circ_* lock the context their circuit was created on, sot_alsz_ctx_* their own context.
 *  - the library links the HIP runtime (libamdhip64). A process that also loads another copy of that runtime -- PyTorch
 *    ships its own -- must load that one first; the first runtime loaded serves every later user, and the other order
 *    leaves the second user without devices ("No HIP GPUs"). flute_amd/lib.py imports torch before the library for this.
 *  - cites are relative to /root/reference/crates.
 */
#ifndef FLUTE_SOT_H
#define FLUTE_SOT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOT_OK 0
#define SOT_EINVAL -1  /* bad argument / shape mismatch */
#define SOT_EDEVICE -2 /* HIP runtime error or no gfx950 device */
#define SOT_ENOMEM -3
#define SOT_ESTATE -4 /* stage called out of order (e.g. hash before encode) */

#define SOT_ROLE_SENDER 0
#define SOT_ROLE_RECEIVER 1

/* QuasiCyclicConf (zappot/src/silent_ot/quasi_cyclic_encode.rs:156-233) + PprfConfig (pprf.rs:40-44) */
typedef struct sot_conf {
    uint64_t n;        /* requested_num_ots */
    uint64_t scaler;   /* MultType::QuasiCyclic{scaler}: 2 (what bench.rs:85 uses) .. 8; scaler - 1 public polynomials a_s */
    uint64_t P;        /* next_prime(max(n, 128*128)) */
    uint64_t parts;    /* num_partitions = GGM trees */
    uint64_t size_per; /* PPRF domain */
    uint64_t N2;
    uint64_t N;
    uint64_t depth;
    uint64_t n_blocks; /* N / 128 */
    uint64_t n64;      /* 2 * n_blocks */
    uint64_t cols;     /* columns of the 128 x cols rT matrix, pprf.rs:91 */
    uint64_t base_ot_count;
} sot_conf;

typedef struct sot_ctx sot_ctx;

const char *sot_last_error(void);
/* number of usable gfx950 devices (0 if none) */
int sot_device_count(void);

/* QuasiCyclicConf::configure(num_ots, scaler, sec_param) -- quasi_cyclic_encode.rs:177-193,
 * silent_ot/mod.rs:671-695. Pure host arithmetic (same f64 operations). */
int sot_configure(uint64_t n, uint64_t scaler, uint64_t sec_param, sot_conf *out);

/* Encoder::configure + pprf::{Sender,Receiver}::new + init_a_polynomials
 * (silent_ot/mod.rs:148,331; quasi_cyclic_encode.rs:127-138): allocates every device buffer for one
 * sender and one receiver instance of `n` OTs and transforms the scaler - 1 public polynomials a_s once. A geometry with
 * n_blocks * 128 < P (about one n in forty at scaler 2, more at larger scalers: N = N2 / scaler is no multiple of 128 and the
 * last, partial Block would be needed to hold P bits; the reference's modp panics on it) is refused with SOT_EINVAL. */
int sot_ctx_create(uint64_t n, uint64_t scaler, int device, sot_ctx **out);
void sot_ctx_destroy(sot_ctx *ctx);
int sot_ctx_conf(const sot_ctx *ctx, sot_conf *out);
/* device bytes held by the context */
uint64_t sot_ctx_device_bytes(const sot_ctx *ctx);
/* block until all work queued by this context has finished */
int sot_ctx_sync(sot_ctx *ctx);
/* the hipStream_t this context launches on (for external event timing) */
void *sot_ctx_stream(sot_ctx *ctx);

/* Stream-ordered use (no counterpart in the reference, whose stages are blocking calls on rayon pools). By default a
 * context owns a stream and every stage call waits for it before returning. sot_ctx_set_stream(ctx, s) makes the context
 * queue its kernels and copies on the caller's hipStream_t s (NULL: back to its own) -- e.g. the stream a collective
 * library orders its work against --, and sot_ctx_set_async(ctx, 1) lets the stage calls that hand nothing to the host
 * (receiver expand, encode, the sot_shard_* calls, hashes / MT bits without host outputs, the *_to_device copies) return
 * without waiting: consecutive stages and the collectives between them are then ordered by the stream alone, and a step
 * of a sharded expansion has one host synchronisation (the TreeGrp download of sot_sender_expand) instead of eight.
 * Small host inputs are copied through a pinned ring inside the call, so the caller's buffers are free on return. Calls
 * that return data to the host still wait. sot_ctx_sync waits for everything queued. */
int sot_ctx_set_stream(sot_ctx *ctx, void *hip_stream);
int sot_ctx_set_async(sot_ctx *ctx, int on);

/* ---- stage 1: PPRF expansion -------------------------------------------------------------- */
/* TreeGrp flat layout (pprf.rs:752-757), group g/8 = 0 .. parts/8-1:
 *   tg_sums [grp][keep(2)][d (depth-1)][tree(8)] Blocks,  tg_last [grp][tree(8)][4] Blocks.
 * The Rust shim (de)serialises these into pprf::Msg::TreeGrp for mpc_channel. */
size_t sot_treegrp_sums_blocks(const sot_conf *c);
size_t sot_treegrp_last_blocks(const sot_conf *c);

/* pprf::Sender::expand(value = delta, InterleavedTransposed) -- pprf.rs:71-281.
 * base_ots: [tree*depth + d][2] Blocks (Array2 order, pprf.rs:67); seed: the Block the reference
 * draws with rng.gen() at pprf.rs:96; num_threads: rayon pool size (roots depend on it, pprf.rs:115).
 * Writes the TreeGrp payload for the peer to host memory. The expanded leaves stay in HBM. */
int sot_sender_expand(sot_ctx *ctx, const void *base_ots, const void *delta, const void *seed, uint64_t num_threads,
                      void *tg_sums_out, void *tg_last_out);
/* pprf::Receiver::expand -- pprf.rs:295-560. base_ots [tree*depth + d] Blocks, choices u8 0/1 same order. */
int sot_receiver_expand(sot_ctx *ctx, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                        const void *tg_last);
/* pprf::Receiver::get_points(InterleavedTransposed) -- pprf.rs:563-577 (host arithmetic) */
int sot_receiver_points(const sot_conf *c, const uint8_t *choices, uint64_t *points_out);
/* The reference's return value of expand: Array2<Block>(128, cols), row major, i.e. the leaves
 * after copy_out's 128x128 bit transposes (pprf.rs:790-835, util/transpose.rs:53-109). */
int sot_download_rT(sot_ctx *ctx, int role, void *rT_out);

/* ---- stage 2: QuasiCyclic encode ---------------------------------------------------------- */
/* QuasiCyclicEncoder::dual_encode (sender, quasi_cyclic_encode.rs:29-52) /
 * dual_encode2 with ChoiceBitPacking::True (receiver, :54-124, silent_ot/mod.rs:552-575).
 * For the receiver `points` are the `parts` interleaved noisy positions S. */
int sot_qc_encode(sot_ctx *ctx, int role, const uint64_t *points, uint64_t npoints);
/* QuasiCyclicEncoder::dual_encode2(rT, sb, choice_bit_packing) for the receiver with the packing as an argument
 * (quasi_cyclic_encode.rs:54-124; silent_ot/mod.rs:410-435). packing != 0: ChoiceBitPacking::True, same as
 * sot_qc_encode(ctx, SOT_ROLE_RECEIVER, ...), C_out is ignored. packing == 0: ChoiceBitPacking::False (what the
 * reference's correlated_silent_ot test uses, silent_ot/mod.rs:867): all 128 rows of A come from rT and the choice
 * vector is returned separately, C_out[i] = 0/1 for i < n (the Option<Vec<u8>> of :104-117). After an unpacked encode
 * sot_receiver_hash / sot_mt_bits return SOT_ESTATE (Receiver::hash takes the choice from bit 0 of A). */
int sot_qc_encode2(sot_ctx *ctx, const uint64_t *points, uint64_t npoints, int packing, uint8_t *C_out);
/* correlated_silent_send / correlated_silent_receive outputs: n Blocks (silent_ot/mod.rs:217,410) */
int sot_download_encoded(sot_ctx *ctx, int role, void *blocks_out);

/* ---- stage 3: hashes + MT bits ------------------------------------------------------------- */
/* Sender::hash (silent_ot/mod.rs:275-287): out = n x [Block;2] */
int sot_sender_hash(sot_ctx *ctx, const void *delta, void *msgs_out);
/* Receiver::hash (silent_ot/mod.rs:470-482): msgs n Blocks, choices BitVec<usize,Lsb0> as ceil(n/64) u64 */
int sot_receiver_hash(sot_ctx *ctx, void *msgs_out, uint64_t *choice_words_out);
/* SilentMtProvider::precompute_mts bit derivation (gmw/src/mul_triple/silent_ot.rs:93-111), fused with
 * both hashes: needs the sender AND receiver instance of this context encoded. a, b, c: ceil(n/64) u64
 * each (MulTriples' three BitVec<usize, Lsb0>, gmw/src/mul_triple/mod.rs:30-35). Host pointers may be
 * NULL to leave the bits on the device only. */
int sot_mt_bits(sot_ctx *ctx, const void *delta, uint64_t *a_out, uint64_t *b_out, uint64_t *c_out);

/* The MT bits of ONE direction, for a party that runs its two directions on two GPUs (role split; the reference runs them
 * on two thread pools and joins, gmw/src/mul_triple/silent_ot.rs:87-111). role = SOT_ROLE_SENDER: b | v with b = lsb(m0) ^
 * lsb(m1), v = lsb(m0) of this direction's messages (delta required); SOT_ROLE_RECEIVER: a | u with a = the choice bit, u =
 * the lsb of its message. 2 * ceil(n / 64) words into a caller-owned DEVICE buffer; the join on any GPU that holds both is
 * a, b, c = a & b ^ u ^ v word by word (flute_amd/dist.py::RoleSplitPipeline gathers them with one RCCL all-gather). */
int sot_mt_role_bits_to_device(sot_ctx *ctx, int role, const void *delta, void *dst_device, uint64_t dst_bytes);
/* The three MT bit vectors of the last sot_mt_bits, a | b | c (3 * ceil(n/64) u64), copied device-to-device into a
 * caller-owned DEVICE buffer on the context's GPU (synchronous on return). No counterpart in the reference (its
 * MulTriples live in host memory, gmw/src/mul_triple/mod.rs:30-35): this is what a multi-GPU party hands to the
 * RCCL all-gather that reassembles its MT store from the per-GPU instances. */
int sot_mt_bits_to_device(sot_ctx *ctx, void *dst_device, uint64_t dst_bytes);
/* The Block outputs of this context -- of this rank's n_local outputs on a sharded one -- written straight into a
 * caller-owned DEVICE buffer on the context's GPU, synchronous on return. what = SOT_OUT_B: the sender's encoded vector B
 * (correlated_silent_send, silent_ot/mod.rs:217), n_out Blocks; SOT_OUT_A: the receiver's A (:410); SOT_OUT_SENDER_MSGS:
 * Sender::hash (:275-287), n_out x [Block; 2], needs delta; SOT_OUT_RECEIVER_MSGS: Receiver::hash (:470-482), n_out Blocks
 * (the choice words stay in the context: sot_receiver_hash with msgs_out = NULL returns them). The reference returns these
 * vectors whole in host memory (:193-211, :391-404); on a sharded context this is what the RCCL all-gather that
 * reassembles them reads (flute_amd/dist.py::gather_sharded_blocks; north_star: "all-gather over xGMI reassembles the
 * output vector"). */
#define SOT_OUT_B 0
#define SOT_OUT_A 1
#define SOT_OUT_SENDER_MSGS 2
#define SOT_OUT_RECEIVER_MSGS 3
int sot_blocks_to_device(sot_ctx *ctx, int what, const void *delta, void *dst_device, uint64_t dst_bytes);

/* ---- one expansion sharded over the GPUs of a node (no counterpart in the reference, whose parallelism is
 * rayon threads: tree groups pprf.rs:128, bit rows quasi_cyclic_encode.rs:35-46, column tiles :140-151) -------------
 * One process per GPU, `world` in {2, 4, 8}. Rank r expands the tree sets [set0, set0 + nsets_local) (or all of them
 * when ggm_replicated != 0), encodes bit rows [r * 128 / world, (r + 1) * 128 / world) of BOTH roles in one pass and
 * hashes the outputs [k0, k0 + n_local). The library moves no bytes between GPUs; the host runs two all-to-alls (RCCL)
 * on DEVICE buffers between the calls below, peer order = rank order, equal counts per peer:
 *     sot_sender_expand + sot_receiver_expand          (own tree sets; TreeGrp arrays: only this rank's groups are
 *                                                       written / read, the flat layout is unchanged)
 *     sot_shard_pack_leaves(send)   -> all-to-all, x1_bytes_per_peer to each peer -> sot_shard_unpack_leaves(recv)
 *       (ggm_replicated: sot_shard_local_leaves instead, no exchange)
 *     sot_shard_qc_encode(points, send2)  -> all-to-all, x2_bytes_per_peer to each peer -> sot_shard_unpack_encoded(recv2)
 *     sot_sender_hash / sot_receiver_hash / sot_mt_bits / sot_download_encoded / sot_mt_bits_to_device / sot_and_circ_*
 *       now work on this rank's n_local outputs; concatenated in rank order (MT words: chunk / 64 words per rank and
 *       vector) they are bit-identical to the one-GPU context's results for the same inputs. */
typedef struct sot_shard_info {
    uint32_t rank, world;
    uint32_t set0, nsets_local; /* tree sets (groups of 8 trees) this rank expands */
    uint32_t ggm_replicated;
    uint32_t rows_per_rank;     /* 128 / world */
    uint64_t k0, n_local;       /* outputs of the hash stage owned by this rank */
    uint64_t chunk;             /* outputs per rank (multiple of 64); k0 = rank * chunk */
    uint64_t x1_bytes_per_peer; /* exchange #1 message size; send / recv buffers hold world of them */
    uint64_t x2_bytes_per_peer; /* exchange #2 message size */
} sot_shard_info;
int sot_ctx_create_sharded(uint64_t n, uint64_t scaler, int device, uint32_t rank, uint32_t world, int ggm_replicated,
                           sot_ctx **out);
int sot_shard_info_get(const sot_ctx *ctx, sot_shard_info *out);
/* leaf slices of this rank's tiles for every peer -> send_dev (device, world * x1_bytes_per_peer); synchronous */
int sot_shard_pack_leaves(sot_ctx *ctx, void *send_dev);
/* the same for one role's half of every message: roles = 1 the sender's leaves (bytes [0, x1 / 2) of each peer's message),
 * 2 the receiver's (bytes [x1 / 2, x1)), 3 both -- so that the sender's halves can be exchanged while the receiver's
 * trees still expand (flute_amd/dist.py::ShardedPipeline, overlap = True) */
int sot_shard_pack_leaves_role(sot_ctx *ctx, void *send_dev, uint32_t roles);
/* recv_dev (device, world * x1_bytes_per_peer, message of rank r at r * x1_bytes_per_peer) -> slot Blocks */
int sot_shard_unpack_leaves(sot_ctx *ctx, const void *recv_dev);
/* ggm_replicated: cut this rank's rows out of the locally expanded leaves */
int sot_shard_local_leaves(sot_ctx *ctx);
/* dual_encode + dual_encode2(packed) of this rank's bit rows of both roles -> send_dev (device,
 * world * x2_bytes_per_peer): the send buffer of exchange #2; synchronous. points / npoints (the `parts` noisy positions
 * S of the receiver) are read on rank 0 only -- the noise vector replaces the receiver's bit row 0, which rank 0 owns --
 * and may be NULL / 0 on the other ranks. */
int sot_shard_qc_encode(sot_ctx *ctx, const uint64_t *points, uint64_t npoints, void *send_dev);
/* recv_dev (device, world * x2_bytes_per_peer) -> this rank's n_local whole Blocks B and A */
int sot_shard_unpack_encoded(sot_ctx *ctx, const void *recv_dev);

/* ---- FLUTE function-dependent setup: AND layers over the expanded triples ------------------ */
/* LutSetupProvider::setup (gmw/src/protocols/aby2_lut.rs:699-783) evaluates, with BooleanGmw and multiplication
 * triples, a circuit that has one AND per subset (size >= 2) of every LUT gate's private input masks
 * (setup_data_circ, :321-370). The circuit is built by the host (graph walk; flute_amd/lut_setup.py mirrors it) and
 * handed over flat: value ids 0 .. n_inputs-1 are the input bits, n_inputs + g is AND gate g; the gates are listed
 * in EXECUTION order = sorted by (layer, gate id), the order inside a layer of the reference (base_circuit.rs:481-484);
 * layer_counts[l] gates belong to layer l; x_src / y_src are the value ids and::compute_shares(x, y, mt) receives
 * (boolean_gmw.rs:54-61; note parent_gates() reverses the wiring order, base_circuit.rs:122-133). Triples are taken
 * from the END of the context's MT store in gate order (Executor::execute, executor.rs:139: split_off_last). */
typedef struct sot_and_circ sot_and_circ;
int sot_and_circ_create(sot_ctx *ctx, uint32_t n_inputs, uint32_t n_gates, uint32_t n_layers,
                        const uint32_t *layer_counts, const uint32_t *x_src, const uint32_t *y_src, sot_and_circ **out);
void sot_and_circ_destroy(sot_and_circ *circ);
/* circ_inputs of setup() (aby2_lut.rs:710,727-733): this party's private share bits, Lsb0 in u64 words */
int sot_and_circ_set_inputs(sot_and_circ *circ, const uint64_t *input_words);
/* BooleanGmw::compute_msg for one layer (boolean_gmw.rs:40-67): Msg::AndLayer { d, e } as the ceil(count/8) bytes
 * of BitVec<u8, Lsb0>::into_vec() (dead bits zero); NULL leaves the message on the device. Consumes the triples. */
int sot_and_circ_layer_msg(sot_and_circ *circ, uint32_t layer, uint8_t *d_out, uint8_t *e_out);
/* BooleanGmw::evaluate_interactive (boolean_gmw.rs:69-95, evaluate.rs:12-21) with the peer's message */
int sot_and_circ_layer_eval(sot_and_circ *circ, uint32_t layer, uint32_t party_id, const uint8_t *d_peer,
                            const uint8_t *e_peer);
/* same, the peer's message read device-to-device from the peer's circuit on the SAME GPU (in-process two-party
 * runs: tests, benchmarks; the reference's analogue is mpc_channel::in_memory) */
int sot_and_circ_layer_eval_peer(sot_and_circ *circ, uint32_t layer, uint32_t party_id, sot_and_circ *peer);
/* evaluate_interactive of `layer` with the peer's message AND compute_msg of layer + 1 in the same call: one host round
 * trip per network round instead of two (the next layer's message depends only on this layer's outputs, executor.rs:139-163).
 * d_next_out / e_next_out: the message of layer + 1 (ignored for the last layer; NULL leaves it on the device). */
int sot_and_circ_layer_eval_msg(sot_and_circ *circ, uint32_t layer, uint32_t party_id, const uint8_t *d_peer,
                                const uint8_t *e_peer, uint8_t *d_next_out, uint8_t *e_next_out);
/* In-process two-party runs (tests, benchmarks; the reference's analogue is two Executors over mpc_channel::in_memory,
 * executor.rs:74-193): layers first .. last of BOTH parties -- circ0 = party 0, circ1 = party 1, same circuit shape, two
 * contexts on the same GPU -- in one call: one fused launch per layer (both parties' compute_msg + evaluate_interactive),
 * one host synchronisation at the end. Each party's triples come from the end of its own context's store. */
int sot_and_circ_run_layers_peer(sot_and_circ *circ0, sot_and_circ *circ1, uint32_t first, uint32_t last);
/* gate_outputs()[0] of the executor (aby2_lut.rs:761): one byte 0/1 per value, inputs then gates in execution order */
int sot_and_circ_values(sot_and_circ *circ, uint8_t *bits_out);
/* FLUTE ONLINE LUT evaluation: LutGate::compute_delta_share (aby2_lut.rs:209-277, expand :947-964) for
 * `batch` SIMD instances of a template of n_luts LUT gates, on the subset products that the evaluated setup circuit
 * `circ` (of N x the template's setup circuit, instance-major) holds in HBM: instances first_instance .. first_instance +
 * batch - 1 of it (one call per LUT layer of a deeper circuit whose layers were set up together). arity[g] = inputs of gate g (1 .. 8);
 * tt[g][8]: its truth table, bit (j & 31) of word j >> 5 = output for row j (input i = bit arity-1-i of j, the reference's
 * row order); lsrc[g][256]: for every non-empty subset s of the row bits where this party's share of the product of those
 * inputs' lambdas lives: i for setup input i of the instance (one input), (l + 1) << 24 | r for the gate of rank r among
 * the template's setup gates of layer l; tcount[l]: template gates per setup layer; masked_in[inst][g]: the public masked
 * inputs as a row index; lam_out[inst][g]: this party's share of the output lambda (0/1). delta_share_out[inst][g] (0/1):
 * the bits of Msg::Delta; the public output is the XOR of both parties' (:145-165). NOTE: the reference's own AND-fold
 * (:1177-1188) starts from all-false and zeroes every term (its tests are #[ignore]d); this computes the protocol as
 * specified -- parity unpinned, checked against the plaintext LUT value instead (tests/test_gpu_lut_eval.py).
 * The device copies of the template tables and of the per-item arrays live in `circ` and are grown on demand: calls
 * with the same or a smaller batch allocate nothing. */
int sot_lut_online_eval(sot_and_circ *circ, uint32_t first_instance, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                        const uint32_t *arity, const uint32_t *tt, const uint32_t *lsrc, const uint32_t *tcount,
                        const uint8_t *masked_in, const uint8_t *lam_out, uint32_t party_id, uint8_t *delta_share_out);
/* sot_lut_online_eval with masked_in, lam_out and delta_share_out (batch x n_luts bytes each) in DEVICE memory on the
 * circuit's GPU; the template tables are uploaded only when they differ from the previous call's. Queued on the context's
 * stream and not waited for: sot_ctx_sync before the result is read. */
int sot_lut_online_eval_dev(sot_and_circ *circ, uint32_t first_instance, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                            const uint32_t *arity, const uint32_t *tt, const uint32_t *lsrc, const uint32_t *tcount,
                            const void *masked_in_dev, const void *lam_out_dev, uint32_t party_id, void *delta_share_dev);
/* triples left in the context's store (MulTriples::len after the split_off_last calls so far) */
uint64_t sot_mt_remaining(const sot_ctx *ctx);

/* ---- ALSZ13 OT extension: zappot::ot_ext after the 128 base OTs ---------------------------------- */
/* Receiver::receive_random, local part (zappot/src/ot_ext/mod.rs:193-235). base_ots: [128][2] Blocks (the base-OT
 * sender's message pairs); choice_bytes: count/8 bytes, Lsb0 (BitSlice<usize> raw bytes); count % 8 == 0.
 * u_rows_out: [128][count/8] bytes, row idx = the ExtOTMsg::URow(idx, row) payloads; msgs_out: count Blocks. */
int sot_alsz_receive(int device, const void *base_ots, const uint8_t *choice_bytes, uint64_t count, uint8_t *u_rows_out,
                     void *msgs_out);
/* Sender::send_random, local part (:91-140). base_ots: [128] Blocks (the base-OT receiver's outputs); base_choices: the
 * 128 base choices as 16 Lsb0 bytes (= delta, :88-90); u_rows: the peer's [128][count/8] bytes. msgs_out: count x
 * [Block; 2]. */
int sot_alsz_send(int device, const void *base_ots, const void *base_choices, const uint8_t *u_rows, uint64_t count,
                  void *msgs_out);

/* Device-resident OT-extension context: one zappot::ot_ext::Sender or Receiver instance (ot_ext/mod.rs:28-46, :144-154)
 * whose device buffers (base-OT keys, choices, the 128 x count u-row matrix, the hashed messages) are allocated once,
 * for up to max_count extensions per call. The one-shot sot_alsz_receive / sot_alsz_send above are this context created
 * and destroyed around a single call. Calls on one context are serialised by a lock; they run on the context's own
 * (non-blocking) stream. ORDERING of device-resident hand-offs: sot_alsz_ctx_send given the sot_alsz_ctx_u_dev pointer
 * of a live receiver context on the same GPU waits (stream-side, hipStreamWaitEvent) for that context's last call, so
 * receive -> send needs no sync in between; any OTHER device pointer handed in (u_dev of the caller's own making,
 * choice_bytes with choices_on_device) must be complete before the call, and before the pointers returned by
 * sot_alsz_ctx_msgs_dev / sot_alsz_ctx_u_dev are read by anything else the producing context must be synced with
 * sot_alsz_ctx_sync. */
typedef struct sot_alsz_ctx sot_alsz_ctx;
int sot_alsz_ctx_create(int device, uint64_t max_count, sot_alsz_ctx **out);
void sot_alsz_ctx_destroy(sot_alsz_ctx *actx);
/* Receiver::receive_random, local part (:193-235). choice_bytes: count/8 bytes in host memory, or in device memory when
 * choices_on_device != 0. u_rows_out ([128][count/8] bytes) and msgs_out (count Blocks) are host buffers or NULL: with
 * both NULL nothing leaves the GPU and the call returns without waiting (sot_alsz_ctx_sync). */
int sot_alsz_ctx_receive(sot_alsz_ctx *actx, const void *base_ots, const uint8_t *choice_bytes, int choices_on_device,
                         uint64_t count, uint8_t *u_rows_out, void *msgs_out);
/* Sender::send_random, local part (:91-140). The receiver's u rows come from host memory (u_rows: [128][count/8] bytes)
 * or, with u_rows == NULL, from device memory (u_dev: [128][ceil(count/128)] Blocks, the layout of sot_alsz_ctx_u_dev of
 * a receiver context on the same GPU: in-process two-party runs). msgs_out: count x [Block; 2] on the host, or NULL. */
int sot_alsz_ctx_send(sot_alsz_ctx *actx, const void *base_ots, const void *base_choices, const uint8_t *u_rows,
                      const void *u_dev, uint64_t count, void *msgs_out);
/* device pointers to the last call's results: messages (count Blocks / count x [Block; 2]) and the u-row matrix */
void *sot_alsz_ctx_msgs_dev(sot_alsz_ctx *actx);
void *sot_alsz_ctx_u_dev(sot_alsz_ctx *actx);
int sot_alsz_ctx_sync(sot_alsz_ctx *actx);

/* The one entry point that allocates nothing. NO counterpart in the reference: it exists for callers that cannot call
 * hipMalloc -- tight serving loops, harnesses that forbid allocation. It is synchronous on the null stream (blocking
 * copies), so it is not for use inside a HIP stream capture. Same arguments,
 * computation and results as sot_alsz_receive, but the device buffers (base-OT keys, padded choices, u-row matrix,
 * hashed messages) are one file-scope device array of 2 150 400 bytes per GPU, held until the process ends; the call
 * makes no hipMalloc / hipFree and no host heap allocation. count <= sot_alsz_noalloc_max_count() = 65 536 OTs per call,
 * larger counts return SOT_EINVAL. Every call on a GPU shares that GPU's array (the last call's contents are
 * overwritten): calls in this process are serialised by a lock; do not rely on anything else for overlapping calls. */
int sot_alsz_receive_noalloc(int device, const void *base_ots, const uint8_t *choice_bytes, uint64_t count,
                             uint8_t *u_rows_out, void *msgs_out);
uint64_t sot_alsz_noalloc_max_count(void);

/* ---- whole-direction pipelines (random_silent_send / random_silent_receive) --------------- */
/* silent_ot/mod.rs:193-211. msgs_out may be NULL (results stay in HBM). */
int sot_random_silent_send(sot_ctx *ctx, const void *base_ots, const void *delta, const void *seed,
                           uint64_t num_threads, void *tg_sums_out, void *tg_last_out, void *msgs_out);
/* silent_ot/mod.rs:391-404 */
int sot_random_silent_receive(sot_ctx *ctx, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                              const void *tg_last, void *msgs_out, uint64_t *choice_words_out);

/* ---- measurement hooks (no counterpart in the reference; bench.rs:206-219 only wall-clocks phases) ------ */
/* Per-kernel HIP-event timing on the context's stream. While enabled every kernel launch is bracketed
 * by two events; sot_prof_read synchronises and returns the accumulated device time and launch count. */
int sot_prof_enable(sot_ctx *ctx, int on);
int sot_prof_reset(sot_ctx *ctx);
int sot_prof_kernels(void);
const char *sot_prof_name(int kernel_id);
int sot_prof_read(sot_ctx *ctx, int kernel_id, double *total_ms, uint64_t *launches);

/* ---- fixture-level entry points ------------------------------------------------------------- */
/* util/transpose.rs:9-14 transpose_128 on `ntiles` independent 128x128 bit tiles (host in/out) */
int sot_transpose128(int device, const void *tiles_in, void *tiles_out, uint64_t ntiles);
/* AES-128 ECB of host blocks with an arbitrary key (crate aes 0.8.3, used for KATs) */
int sot_aes128_encrypt(int device, const void *key16, const void *in, void *out, uint64_t nblocks);
/* AesHash::cr_hash_slice_mut (util/aes_hash.rs:47-55) on host blocks */
int sot_cr_hash(int device, const void *in, void *out, uint64_t nblocks);
/* QuasiCyclicEncoder::dual_encode / dual_encode2(packed) on a caller-supplied rT
 * (Array2<Block>(128, cols), host) -- quasi_cyclic_encode.rs:29-124. Un-transposes rT on the device and
 * runs the same encode kernels; result (n Blocks) is written to blocks_out and kept in HBM for the hashes. */
int sot_qc_dual_encode_host(sot_ctx *ctx, int role, const void *rT, const uint64_t *points, uint64_t npoints,
                            void *blocks_out);
/* Validation hook for the product inside sot_qc_encode (no reference counterpart: bitpolymul's transform,
 * bitpolymul2/bitpolymul.c:168-279, is replaced by a Frobenius-pruned additive FFT over GF(2^32)).
 * Runs the forward transform stages on ONE bit row given by the host (2^(m-6) words = the low half of the
 * 2^m-bit capacity, m = sot_qc_log2_capacity) and returns the 2^(m-5) words of that row after
 * stage 0: basis conversion, 1: + levels 1..8, 2: + levels 9..12, 3: + levels 13..m-9, 4: all levels.
 * tests compare the result with the executable model flute_amd/csrc/fafft_model.py. */
int sot_qc_transform_row(sot_ctx *ctx, const uint32_t *row_in, int stage, uint32_t *row_out);
uint32_t sot_qc_log2_capacity(const sot_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
