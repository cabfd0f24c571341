/*
 * sot_oracle.h -- CPU restatement ("oracle") of the FLUTE / zappot silent-OT
 * expansion hot path.
 *
 * THIS IS TEST INFRASTRUCTURE. It is the checker for the HIP path, never the
 * product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it. The product library (flute_amd/csrc) does not link it.
 *
 * Every function cites the reference file:line (relative to
 * /root/reference/crates) whose behaviour it restates.
 *
 * Parity status: the reference is Rust and cannot be built in this image
 * (no cargo/rustc), so byte-for-byte equality with the Rust binary is pinned
 * by (a) the reference's own known-answer tests (basic_bit_shift_xor,
 * sse_rust 16x16 transpose, basic_modp), (b) FIPS-197 AES vectors, (c) the
 * reference's property tests restated in tests/ (silent_pprf,
 * check_correlated, check_random, MT relation) and (d) the reference's own C
 * multiplier bitpolymul2 compiled from its sources into oracle/_ref.
 * GGM / QC / hash outputs beyond that are "parity unpinned" by the
 * reference's tests.
 */
#ifndef SOT_ORACLE_H
#define SOT_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* zappot/src/util/block.rs:17-21 -- Block = u128, little endian.
 * w[0] = bits 0..63, w[1] = bits 64..127; bit i of the Block = (data >> i) & 1 */
typedef struct {
    uint64_t w[2];
} orc_block;

/* zappot/src/silent_ot/quasi_cyclic_encode.rs:156-233 + pprf.rs:40-44,623-631 */
typedef struct {
    uint64_t n;        /* requested_num_ots */
    uint64_t scaler;
    uint64_t P;        /* next_prime(max(n, 128*128)) */
    uint64_t parts;    /* num_partitions == pprf pnt_count */
    uint64_t size_per; /* pprf domain */
    uint64_t N2;
    uint64_t N;
    uint64_t depth;    /* log2_ceil(size_per) */
    uint64_t n_blocks; /* N / 128 */
    uint64_t n64;      /* 2 * n_blocks */
    uint64_t cols;     /* (parts*size_per + 127)/128, pprf.rs:91 */
    uint64_t base_ot_count; /* depth * parts */
} orc_conf;

/* ---- AES-128 (crate aes 0.8.3 == FIPS-197) -------------------------- */
int orc_have_aesni(void);
/* force portable (table) code paths for cross checks: 1 = portable only */
void orc_force_portable(int on);
void orc_aes128_encrypt_blocks(const uint8_t key[16], const uint8_t *in, uint8_t *out, size_t nblocks);
/* util/aes_rng.rs:101-136: block j of the stream = AES_seed(LE128(first+j)) */
void orc_aes_rng_blocks(const uint8_t seed[16], uint64_t first_block, uint8_t *out, size_t nblocks);
/* util/aes_hash.rs:47-66: x[i] = AES_kh(x[i]) ^ x[i] */
void orc_cr_hash(orc_block *x, size_t n);

/* ---- configuration --------------------------------------------------- */
uint64_t orc_next_prime(uint64_t x); /* num_prime next_prime: smallest prime > x */
int orc_configure(uint64_t n, uint64_t scaler, uint64_t sec_param, orc_conf *out);

/* ---- PPRF ------------------------------------------------------------ */
/* pprf.rs:701-706 */
uint64_t orc_get_active_path(const uint8_t *choice_bits, uint64_t depth);
/* pprf.rs:716-741 (InterleavedTransposed) */
uint64_t orc_interleave_point(uint64_t point, uint64_t tree_idx, uint64_t total_trees);

/*
 * Flat TreeGrp layout (pprf.rs:752-757), one record per group g/8 = 0..parts/8-1:
 *   sums     [grp][keep(2)][d (depth-1)][tree(8)]   Blocks
 *   last_ots [grp][tree(8)][4]                      Blocks
 */
/* pprf.rs:71-281. base_ots[tree*depth + d][2]; out_rT is 128 x cols, row major. */
int orc_pprf_sender_expand(const orc_conf *c, const orc_block *base_ots, orc_block delta, orc_block seed,
                           uint64_t num_threads, orc_block *out_rT, orc_block *tg_sums, orc_block *tg_last);
/* pprf.rs:295-560. base_ots[tree*depth + d], choices[tree*depth + d] (u8 0/1) */
int orc_pprf_receiver_expand(const orc_conf *c, const orc_block *base_ots, const uint8_t *choices,
                             const orc_block *tg_sums, const orc_block *tg_last, orc_block *out_rT);
/* pprf.rs:563-577: points in InterleavedTransposed format */
void orc_pprf_points(const orc_conf *c, const uint8_t *choices, uint64_t *points_out);

/* ---- transpose --------------------------------------------------------- */
/* util/transpose.rs:53-109 semantics, any rows%8==0, cols%8==0 (naive bit loop) */
void orc_transpose_naive(const uint8_t *in, uint8_t *out, uint64_t nrows, uint64_t ncols);
/* util/transpose.rs:9-14 */
void orc_transpose_128(orc_block m[128]);

/* ---- GF(2)[x] product (bitpolymul semantics, bitpolymul.c:168-279) ---- */
/* c[0..2*n64) = a[0..n64) * b[0..n64), bit k of word k/64 = coeff of x^k */
void orc_gf2x_mul(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n64);
void orc_gf2x_mul_naive(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n64);
/* optional external multiplier (oracle/_ref bitpolymul_2_128); NULL = own Karatsuba */
typedef void (*orc_mul_fn)(uint64_t *c, const uint64_t *a, const uint64_t *b, unsigned n64);
void orc_set_external_mul(orc_mul_fn fn);

/* ---- QuasiCyclic encode ------------------------------------------------ */
/* quasi_cyclic_encode.rs:329-344 */
void orc_bit_shift_xor(orc_block *dest, uint64_t dest_len, const orc_block *inp, uint64_t inp_len, uint8_t bit_shift);
/* quasi_cyclic_encode.rs:289-327 */
void orc_modp(orc_block *dest, uint64_t dest_len, const orc_block *inp, uint64_t inp_len, uint64_t p);
/* quasi_cyclic_encode.rs:127-138: a_s for s=1 (scaler 2): n64 u64 of AesRng(seed=Block(1)) */
void orc_qc_a_poly(const orc_conf *c, uint64_t *a_out);
void orc_qc_a_poly_s(const orc_conf *c, uint64_t sidx, uint64_t *a_out);
/* quasi_cyclic_encode.rs:29-52. rT 128 x cols. out_B has n Blocks. */
int orc_qc_dual_encode(const orc_conf *c, const orc_block *rT, orc_block *out_B);
/* quasi_cyclic_encode.rs:54-124 + mod.rs:552-575. S = points (parts entries).
 * packed != 0: ChoiceBitPacking::True (out_C unused). packed == 0: out_C gets n bytes. */
int orc_qc_dual_encode2(const orc_conf *c, const orc_block *rT, const uint64_t *S, uint64_t nS, int packed,
                        orc_block *out_A, uint8_t *out_C);
/* optional: keep c_mod_p1 (128 x n_blocks) of the last encode for fixtures */
const orc_block *orc_last_cmodp(uint64_t *rows, uint64_t *cols);

/* ---- hashes + MTs ------------------------------------------------------ */
/* silent_ot/mod.rs:275-287: out[2*i], out[2*i+1] */
void orc_sender_hash(orc_block delta, const orc_block *B, uint64_t n, orc_block *out);
/* silent_ot/mod.rs:470-482: A hashed in place, choices as packed bits Lsb0 in u64 words */
void orc_receiver_hash(orc_block *A, uint64_t n, uint64_t *choice_words);
/* gmw/src/mul_triple/silent_ot.rs:93-111: a,b,c packed Lsb0 in u64 words (ceil(n/64)) */
void orc_mt_bits(const orc_block *send_msgs /*2n*/, const orc_block *recv_msgs /*n*/, const uint64_t *recv_choice_words,
                 uint64_t n, uint64_t *a, uint64_t *b, uint64_t *c);

/* OpenMP team size for the calling host thread (per-thread); the GGM workers, the 128 QC rows and the output
 * transposes run on that team */
void orc_set_num_threads(int n);
int orc_get_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
