// Internal declarations shared by the HIP translation units of libflute_sot.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/flute_sot.h"

// 16-byte Block as four little-endian 32-bit words (word 0 = bits 0..31)
typedef uint4 blk_t;

void sot_set_error(const char *fmt, ...);

#define SOT_HIP(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            sot_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return SOT_EDEVICE;                                                                 \
        }                                                                                       \
    } while (0)

// end of a stage call that hands nothing to the host: wait for the stream unless the context is stream-ordered
#define SOT_SYNC(c)                                               \
    do {                                                          \
        if (!(c)->async) SOT_HIP(hipStreamSynchronize((c)->stream)); \
    } while (0)

#define SOT_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != SOT_OK) return _rc; \
    } while (0)

// AES-128 expanded key: 44 little-endian words, then the same words rotated right by 8 bits (aes_dev.h)
struct aes_rk {
    uint32_t w[88];
};
void aes_expand_key_host(const uint8_t key[16], aes_rk *out);

// geometry of the GF(2^32) Frobenius additive-FFT product used by the QC encode (qc.hip)
struct qc_geom {
    uint32_t m;        // log2 of the product capacity in bits (>= 21)
    uint64_t S;        // bits per operand handled by the FFT = 2^(m-1)
    uint64_t tail;     // P - S if P > S else 0 (handled by the schoolbook tail terms)
    uint64_t lo_bits;  // min(P, S)
    uint64_t nw;       // 32-bit words per bit row = 2^(m-5)
};

// kernel ids for the optional per-kernel HIP-event timing (sot_prof_*)
enum sot_kid {
    KID_ROOTS = 0,
    KID_GGM_LEVEL,
    KID_GGM_LEAF,
    KID_SENDER_TG,
    KID_RECV_INIT,
    KID_RECV_FIX,
    KID_RECV_LAST,
    KID_TRANSPOSE,
    KID_SENDER_HASH,
    KID_RECEIVER_HASH,
    KID_MT_BITS,
    KID_CLEAR_LSB,
    KID_QC_LOAD,
    KID_CONV_FWD,
    KID_CONV_FWD_TILE,
    KID_FFT_TOP,
    KID_FFT_MID,
    KID_FFT_LOW_MUL,
    KID_IFFT_MID,
    KID_IFFT_TOP,
    KID_CONV_INV_TILE,
    KID_CONV_INV,
    KID_QC_STORE,
    KID_QC_FINAL,
    KID_AND_MSG,
    KID_AND_EVAL,
    KID_SHARD_PACK,
    KID_SHARD_UNPACK,
    KID_COUNT
};
extern const char *const SOT_KID_NAMES[KID_COUNT];

struct prof_rec {
    int id;
    hipEvent_t a, b;
};

struct role_state {
    blk_t *V = nullptr;        // leaves, interleaved order, cols*128 Blocks (rT before the bit transposes)
    blk_t *enc = nullptr;      // encoded output, n Blocks (B for the sender, A for the receiver)
    bool expanded = false;
    bool encoded = false;
};

// result-preserving kernel-selection switches, read from the environment ONCE at sot_ctx_create (tests compare the
// fused kernels with the generic ones through them; none of them can skip computation)
enum sot_opt {
    SOT_OPT_NO_TOP = 1,      // SOT_NO_TOP: one-phase-per-launch conversion kernels, whole-unit phases outside the top pass
    SOT_OPT_NO_ASPLIT = 2,   // SOT_NO_ASPLIT: Taylor group A as one tile geometry at R >= 11
    SOT_OPT_NO_WU = 4,       // SOT_NO_WU: whole-unit phases through the generic kernel
    SOT_OPT_NO_W32PAIR = 8,  // SOT_NO_W32PAIR: extra 32-plane passes one row half per workgroup
    SOT_OPT_NO_GGM_TOP = 16,  // SOT_NO_GGM_TOP: every GGM level as its own launch
    SOT_OPT_NO_LUT_COOP = 32  // SOT_NO_LUT_COOP: online LUT evaluation one thread per item (k_lut_online) instead of k_lut_online2
};

// One expansion sharded over `world` GPUs (shard.hip). Rank r owns
//   - the tree sets [set0, set0 + nl) of the GGM stage (unless the GGM stage is replicated),
//   - the bit rows [r * rpr, (r + 1) * rpr), rpr = 128 / world, of BOTH roles' leaf matrices in the QC stage, as "slots"
//     of one combined Block per leaf: bits [0, rpr) = sender rows, bits [rpr, 2 rpr) = receiver rows, the rest zero,
//   - the outputs k in [k0, k0 + n_local) of the hash / MT-bit stage.
struct shard_state {
    uint32_t rank = 0, world = 1;
    uint32_t rpr = 128;          // bit rows per rank and role
    bool ggm_replicated = false;  // every rank expands all trees (no exchange #1) instead of its own tree sets
    uint32_t nl_max = 0;         // tree sets per rank, maximum over ranks
    uint64_t nk = 0;             // leaf tiles per tree set = ceil(cols / nsets)
    uint64_t chunk = 0;          // outputs per rank (a multiple of 64: MT words never straddle ranks)
    uint64_t k0 = 0, n_local = 0;
    uint64_t x1_bytes = 0;       // exchange #1: bytes this rank sends to EACH peer (leaf slices, both roles)
    uint64_t x2_bytes = 0;       // exchange #2: bytes this rank sends to EACH peer (encoded slices of chunk outputs)
    blk_t *W = nullptr;          // slot Blocks of every leaf, cols * 128
    bool slots_ready = false, enc_slots_ready = false;
};

#define QC_NCTR 1024
struct sot_ctx {
    // every sot_* entry point that takes the context holds this lock for the whole call: sender and receiver share the
    // GGM scratch, the Z buffers and one stream, and the reference drives both directions from two threads
    // (gmw/src/mul_triple/silent_ot.rs:87-91). Recursive: the pipeline entry points call the stage entry points.
    std::recursive_mutex mu;
    uint32_t opts = 0;
    int device = 0;
    uint32_t num_cus = 256;  // persistent kernels launch one workgroup per CU
    hipStream_t stream = nullptr;      // the stream every kernel and copy of this context is queued on
    hipStream_t own_stream = nullptr;  // created with the context; `stream` is this one unless sot_ctx_set_stream gave another
    bool async = false;                // sot_ctx_set_async: stage calls that hand nothing to the host do not wait for the stream
    void *h_stage = nullptr;           // pinned staging ring for small host inputs (stage_h2d, async mode)
    uint64_t h_stage_bytes = 0, h_stage_off = 0;
    hipEvent_t stage_done = nullptr;   // recorded behind the last H2D copy out of h_stage
    sot_conf conf;
    qc_geom g;
    uint64_t nsets = 0;  // parts / 8
    uint32_t ggm_set0 = 0, ggm_nsets = 0;  // the tree sets this context expands (all of them unless sharded)
    uint64_t n_out = 0;  // outputs held in role[].enc: conf.n, or this rank's n_local when sharded
    shard_state sh;
    uint64_t dev_bytes = 0;
    role_state role[2];
    // GGM scratch
    blk_t *lvl_a = nullptr, *lvl_b = nullptr;  // ping-pong interior levels
    blk_t *sums = nullptr;                     // [depth][nsets][2][8] Blocks
    blk_t *d_base = nullptr;                   // base OTs (sender: pairs)
    uint8_t *d_choices = nullptr;
    blk_t *d_tg_sums = nullptr, *d_tg_last = nullptr;
    uint32_t *d_rks = nullptr;  // root PRG round keys, [T][88]
    uint64_t rks_cap = 0;
    // QC scratch
    uint32_t *Z0 = nullptr, *Z1 = nullptr;  // bit rows [word][128 rows], ping-pong (the transform runs in place)
    uint32_t rows_active = 128;             // bit rows that hold data (qc_set_rows_active)
    uint32_t *TW = nullptr, *TWC = nullptr; // twiddle matrix tables
    uint32_t *LM = nullptr;                 // per value word packed twiddle masks of the intra-word levels [nw/32][5][240] (packed: see d_LOWCMAP)
    uint32_t *d_ctr = nullptr;               // QC_NCTR zeroed work counters (qc_next_counter)
    uint32_t ctr_next = 0;
    uint32_t *MT1 = nullptr, *MT2 = nullptr; // twiddle masks of the 16-plane passes (levels 9..12 / 13..16)
    void *qc_plan = nullptr;                // fft_plan (qc.hip): tile descriptors of the passes
    uint32_t *up_list = nullptr;            // move lists of the whole-unit conversion phases
    // the public polynomials a_s, s = 1 .. scaler - 1 (quasi_cyclic_encode.rs:127-138), one block of each table per s
    uint32_t *Amat = nullptr;               // [s-1][nw/32][432] per value word multiply-by-FFT(a_s) bit matrices (packed)
    uint32_t *Amat_cur = nullptr;           // the block the low kernel multiplies by (qc_encode_core sets it per pass)
    uint32_t *a2 = nullptr;                 // [s-1][a2_stride] a_s mod (x^P - 1) as a bit vector (+ 2 padding words)
    uint32_t *a_hi = nullptr;               // [s-1][ahi_stride] tail bits a_s'[S .. P)
    uint64_t amat_stride = 0, a2_stride = 0, ahi_stride = 0;  // words
    uint64_t *d_points = nullptr;
    uint32_t *sbv = nullptr;  // receiver noise indicator: one bit per Block of V (silent_ot/mod.rs:560-574)
    // MT bits
    uint64_t *d_mt = nullptr;  // a | b | c words
    uint64_t *d_chw = nullptr; // receiver choice words of sot_receiver_hash (kept apart from the MT store)
    bool recv_unpacked = false;  // the receiver's last encode kept bit row 0 (ChoiceBitPacking::False): A carries no choice bit
    bool mt_valid = false;     // set by hash_mt_bits, cleared by the next expand
    uint64_t mt_rem = 0;       // triples not yet consumed by sot_and_circ_layer_msg (taken from the END of the store)
    void *d_tmp = nullptr;     // download staging
    uint64_t tmp_bytes = 0;
    // profiling
    bool prof = false;
    std::vector<prof_rec> prof_pending;
    std::vector<hipEvent_t> prof_free;
    double prof_ms[KID_COUNT] = {0};
    uint64_t prof_n[KID_COUNT] = {0};
};

void prof_begin(sot_ctx *c, int id);
void prof_end(sot_ctx *c, int id);
void prof_resolve(sot_ctx *c);
#define SOT_LAUNCH(c, id, kernel, grid, block, ...)                              \
    do {                                                                         \
        if ((c)->prof) prof_begin((c), (id));                                    \
        hipLaunchKernelGGL(kernel, grid, block, 0, (c)->stream, __VA_ARGS__);    \
        if ((c)->prof) prof_end((c), (id));                                      \
    } while (0)

// ---- ggm.hip
int ggm_sender_expand(sot_ctx *c, const void *base_ots, const void *delta, const void *seed, uint64_t T,
                      void *tg_sums_out, void *tg_last_out);
int ggm_receiver_expand(sot_ctx *c, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                        const void *tg_last);
int ggm_download_rT(sot_ctx *c, int role, void *out);
int ggm_upload_rT(sot_ctx *c, int role, const void *rT);
// dst_dev: where the hashed messages go on the device (nullptr = the context's staging buffer)
int hash_sender(sot_ctx *c, const void *delta, void *msgs_out, blk_t *dst_dev = nullptr);
int hash_receiver(sot_ctx *c, void *msgs_out, uint64_t *choices_out, blk_t *dst_dev = nullptr);
int hash_mt_bits(sot_ctx *c, const void *delta, uint64_t *a, uint64_t *b, uint64_t *cc);
int hash_mt_role_bits(sot_ctx *c, int role, const void *delta, uint64_t *dst_device);
int util_transpose128(int device, const void *in, void *out, uint64_t ntiles);
int util_aes(int device, const void *key16, const void *in, void *out, uint64_t n, int cr_hash);
int util_aes_ctr_device(sot_ctx *c, const uint8_t key16[16], uint64_t first, blk_t *d_out, uint64_t nblocks);

// ---- qc.hip
int qc_init(sot_ctx *c);
void qc_set_rows_active(sot_ctx *c, uint32_t rows);
void qc_free(sot_ctx *c);
int qc_encode(sot_ctx *c, int role, const uint64_t *points, uint64_t npoints, int packed = 1, uint8_t *C_out = nullptr);
// both roles of a sharded context at once: slot Blocks W -> slot-form outputs (the send buffer of exchange #2)
int qc_encode_slots(sot_ctx *c, const uint64_t *points, uint64_t npoints, blk_t *out);
int qc_debug_row(sot_ctx *c, const uint32_t *row_in, int stage, uint32_t *row_out);

int dev_alloc(sot_ctx *c, void **p, uint64_t bytes);
// small host input -> device on the context's stream. Stream-ordered contexts (sot_ctx_set_async) copy it through the
// context's pinned ring first, so that the caller's buffer is free when the call returns although nothing waits for the GPU.
int stage_h2d(sot_ctx *c, void *dst_dev, const void *src_host, uint64_t bytes);

// ---- shard.hip
int shard_init(sot_ctx *c, uint32_t rank, uint32_t world, int ggm_replicated);
void shard_free(sot_ctx *c);
// set range of rank r when `nsets` tree sets are dealt to `world` ranks in contiguous runs
static inline void shard_sets_of(uint32_t nsets, uint32_t world, uint32_t r, uint32_t *set0, uint32_t *nl) {
    const uint32_t base = nsets / world, rem = nsets % world;
    *set0 = r * base + (r < rem ? r : rem);
    *nl = base + (r < rem ? 1u : 0u);
}
