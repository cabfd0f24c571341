/*
 * sot_oracle.c -- CPU restatement of the zappot silent-OT expansion path.
 * TEST INFRASTRUCTURE ONLY (see sot_oracle.h). Cites are relative to
 * /root/reference/crates.
 */
#include "sot_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_X86 1
#else
#define ORC_X86 0
#endif

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* Block helpers (util/block.rs)                                       */
/* ------------------------------------------------------------------ */
static inline orc_block bxor(orc_block a, orc_block b) {
    orc_block r = {{a.w[0] ^ b.w[0], a.w[1] ^ b.w[1]}};
    return r;
}
static inline orc_block bzero_(void) {
    orc_block r = {{0, 0}};
    return r;
}
static inline orc_block bones_(void) {
    orc_block r = {{~0ULL, ~0ULL}};
    return r;
}
/* block.rs:209-233: plain u128 shifts (shift in 0..127) */
static inline orc_block bshr(orc_block a, unsigned s) {
    orc_block r;
    if (s == 0) return a;
    if (s >= 64) {
        r.w[0] = a.w[1] >> (s - 64);
        r.w[1] = 0;
    } else {
        r.w[0] = (a.w[0] >> s) | (a.w[1] << (64 - s));
        r.w[1] = a.w[1] >> s;
    }
    return r;
}
static inline orc_block bshl(orc_block a, unsigned s) {
    orc_block r;
    if (s == 0) return a;
    if (s >= 128) return bzero_();
    if (s >= 64) {
        r.w[1] = a.w[0] << (s - 64);
        r.w[0] = 0;
    } else {
        r.w[1] = (a.w[1] << s) | (a.w[0] >> (64 - s));
        r.w[0] = a.w[0] << s;
    }
    return r;
}

/* ------------------------------------------------------------------ */
/* AES-128, FIPS-197 (third-party crate aes 0.8.3 in the reference)    */
/* ------------------------------------------------------------------ */
static const uint8_t SBOX[256] = {
    0x63, 0x7c, 0x77, 0x7b, 0xf2, 0x6b, 0x6f, 0xc5, 0x30, 0x01, 0x67, 0x2b, 0xfe, 0xd7, 0xab, 0x76, 0xca, 0x82, 0xc9,
    0x7d, 0xfa, 0x59, 0x47, 0xf0, 0xad, 0xd4, 0xa2, 0xaf, 0x9c, 0xa4, 0x72, 0xc0, 0xb7, 0xfd, 0x93, 0x26, 0x36, 0x3f,
    0xf7, 0xcc, 0x34, 0xa5, 0xe5, 0xf1, 0x71, 0xd8, 0x31, 0x15, 0x04, 0xc7, 0x23, 0xc3, 0x18, 0x96, 0x05, 0x9a, 0x07,
    0x12, 0x80, 0xe2, 0xeb, 0x27, 0xb2, 0x75, 0x09, 0x83, 0x2c, 0x1a, 0x1b, 0x6e, 0x5a, 0xa0, 0x52, 0x3b, 0xd6, 0xb3,
    0x29, 0xe3, 0x2f, 0x84, 0x53, 0xd1, 0x00, 0xed, 0x20, 0xfc, 0xb1, 0x5b, 0x6a, 0xcb, 0xbe, 0x39, 0x4a, 0x4c, 0x58,
    0xcf, 0xd0, 0xef, 0xaa, 0xfb, 0x43, 0x4d, 0x33, 0x85, 0x45, 0xf9, 0x02, 0x7f, 0x50, 0x3c, 0x9f, 0xa8, 0x51, 0xa3,
    0x40, 0x8f, 0x92, 0x9d, 0x38, 0xf5, 0xbc, 0xb6, 0xda, 0x21, 0x10, 0xff, 0xf3, 0xd2, 0xcd, 0x0c, 0x13, 0xec, 0x5f,
    0x97, 0x44, 0x17, 0xc4, 0xa7, 0x7e, 0x3d, 0x64, 0x5d, 0x19, 0x73, 0x60, 0x81, 0x4f, 0xdc, 0x22, 0x2a, 0x90, 0x88,
    0x46, 0xee, 0xb8, 0x14, 0xde, 0x5e, 0x0b, 0xdb, 0xe0, 0x32, 0x3a, 0x0a, 0x49, 0x06, 0x24, 0x5c, 0xc2, 0xd3, 0xac,
    0x62, 0x91, 0x95, 0xe4, 0x79, 0xe7, 0xc8, 0x37, 0x6d, 0x8d, 0xd5, 0x4e, 0xa9, 0x6c, 0x56, 0xf4, 0xea, 0x65, 0x7a,
    0xae, 0x08, 0xba, 0x78, 0x25, 0x2e, 0x1c, 0xa6, 0xb4, 0xc6, 0xe8, 0xdd, 0x74, 0x1f, 0x4b, 0xbd, 0x8b, 0x8a, 0x70,
    0x3e, 0xb5, 0x66, 0x48, 0x03, 0xf6, 0x0e, 0x61, 0x35, 0x57, 0xb9, 0x86, 0xc1, 0x1d, 0x9e, 0xe1, 0xf8, 0x98, 0x11,
    0x69, 0xd9, 0x8e, 0x94, 0x9b, 0x1e, 0x87, 0xe9, 0xce, 0x55, 0x28, 0xdf, 0x8c, 0xa1, 0x89, 0x0d, 0xbf, 0xe6, 0x42,
    0x68, 0x41, 0x99, 0x2d, 0x0f, 0xb0, 0x54, 0xbb, 0x16};

/* OpenMP team size of the CALLING thread (a per-thread setting): the CPU baseline runs the sender and the receiver
 * direction from two host threads with available_parallelism() / 2 workers each, gmw/src/mul_triple/silent_ot.rs:47,87-91 */
void orc_set_num_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
int orc_get_max_threads(void) { return omp_get_max_threads(); }

static int g_force_portable = 0;
void orc_force_portable(int on) { g_force_portable = on; }

int orc_have_aesni(void) {
#if ORC_X86
    return !g_force_portable && __builtin_cpu_supports("aes") && __builtin_cpu_supports("sse4.1");
#else
    return 0;
#endif
}
static int have_pclmul(void) {
#if ORC_X86
    return !g_force_portable && __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
#else
    return 0;
#endif
}

static inline uint8_t xtime(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }

/* FIPS-197 key expansion: 11 round keys of 16 bytes */
static void aes_expand_key(const uint8_t key[16], uint8_t rk[176]) {
    static const uint8_t rcon[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36};
    memcpy(rk, key, 16);
    for (int i = 4; i < 44; i++) {
        uint8_t t[4];
        memcpy(t, rk + 4 * (i - 1), 4);
        if (i % 4 == 0) {
            uint8_t u = t[0];
            t[0] = SBOX[t[1]] ^ rcon[i / 4 - 1];
            t[1] = SBOX[t[2]];
            t[2] = SBOX[t[3]];
            t[3] = SBOX[u];
        }
        for (int k = 0; k < 4; k++) rk[4 * i + k] = rk[4 * (i - 4) + k] ^ t[k];
    }
}

static void aes_encrypt_portable(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16]) {
    uint8_t s[16], t[16];
    for (int i = 0; i < 16; i++) s[i] = in[i] ^ rk[i];
    for (int r = 1; r <= 10; r++) {
        /* SubBytes + ShiftRows: state is column major, s[4*c + row] */
        for (int c = 0; c < 4; c++)
            for (int row = 0; row < 4; row++) t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];
        if (r < 10) {
            for (int c = 0; c < 4; c++) {
                uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                uint8_t x = a0 ^ a1 ^ a2 ^ a3;
                s[4 * c + 0] = a0 ^ x ^ xtime(a0 ^ a1);
                s[4 * c + 1] = a1 ^ x ^ xtime(a1 ^ a2);
                s[4 * c + 2] = a2 ^ x ^ xtime(a2 ^ a3);
                s[4 * c + 3] = a3 ^ x ^ xtime(a3 ^ a0);
            }
        } else {
            memcpy(s, t, 16);
        }
        for (int i = 0; i < 16; i++) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

#if ORC_X86
__attribute__((target("aes,sse4.1"))) static void aes_encrypt_ni(const uint8_t rk[176], const uint8_t *in,
                                                                  uint8_t *out, size_t n) {
    __m128i k[11];
    for (int i = 0; i < 11; i++) k[i] = _mm_loadu_si128((const __m128i *)(rk + 16 * i));
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        __m128i b[8];
        for (int j = 0; j < 8; j++) b[j] = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(in + 16 * (i + j))), k[0]);
        for (int r = 1; r < 10; r++)
            for (int j = 0; j < 8; j++) b[j] = _mm_aesenc_si128(b[j], k[r]);
        for (int j = 0; j < 8; j++)
            _mm_storeu_si128((__m128i *)(out + 16 * (i + j)), _mm_aesenclast_si128(b[j], k[10]));
    }
    for (; i < n; i++) {
        __m128i b = _mm_xor_si128(_mm_loadu_si128((const __m128i *)(in + 16 * i)), k[0]);
        for (int r = 1; r < 10; r++) b = _mm_aesenc_si128(b, k[r]);
        _mm_storeu_si128((__m128i *)(out + 16 * i), _mm_aesenclast_si128(b, k[10]));
    }
}
#endif

typedef struct {
    uint8_t rk[176];
} aes_key;

static void aes_init(aes_key *k, const uint8_t key[16]) { aes_expand_key(key, k->rk); }

/* in and out may alias exactly */
static void aes_encrypt_n(const aes_key *k, const uint8_t *in, uint8_t *out, size_t n) {
#if ORC_X86
    if (orc_have_aesni()) {
        aes_encrypt_ni(k->rk, in, out, n);
        return;
    }
#endif
    for (size_t i = 0; i < n; i++) {
        uint8_t tmp[16];
        aes_encrypt_portable(k->rk, in + 16 * i, tmp);
        memcpy(out + 16 * i, tmp, 16);
    }
}

void orc_aes128_encrypt_blocks(const uint8_t key[16], const uint8_t *in, uint8_t *out, size_t nblocks) {
    aes_key k;
    aes_init(&k, key);
    aes_encrypt_n(&k, in, out, nblocks);
}

/* util/aes_rng.rs:109-122: counter blocks state.to_le_bytes(), state += 1 */
void orc_aes_rng_blocks(const uint8_t seed[16], uint64_t first_block, uint8_t *out, size_t nblocks) {
    aes_key k;
    aes_init(&k, seed);
    const size_t CH = 1024;
    for (size_t i = 0; i < nblocks; i += CH) {
        size_t m = nblocks - i < CH ? nblocks - i : CH;
        for (size_t j = 0; j < m; j++) {
            uint64_t ctr = first_block + i + j;
            memcpy(out + 16 * (i + j), &ctr, 8);
            memset(out + 16 * (i + j) + 8, 0, 8);
        }
        aes_encrypt_n(&k, out + 16 * i, out + 16 * i, m);
    }
}

/* u128 constant -> LE key bytes */
static void u128_key(uint64_t lo, uint64_t hi, uint8_t key[16]) {
    memcpy(key, &lo, 8);
    memcpy(key + 8, &hi, 8);
}

/* pprf.rs:675-688: 91389970179024809574621370423327856399 and
 * 297966570818470707816499469807199042980 as u128 LE bytes
 * (= 0f173c46 8f95ba6b 9cac20a8 6911c144 / a4bdc17a 018bc891 6ab3db80 2c462ae0) */
static const uint8_t KEY_GGM0[16] = {0x0f, 0x17, 0x3c, 0x46, 0x8f, 0x95, 0xba, 0x6b,
                                     0x9c, 0xac, 0x20, 0xa8, 0x69, 0x11, 0xc1, 0x44};
static const uint8_t KEY_GGM1[16] = {0xa4, 0xbd, 0xc1, 0x7a, 0x01, 0x8b, 0xc8, 0x91,
                                     0x6a, 0xb3, 0xdb, 0x80, 0x2c, 0x46, 0x2a, 0xe0};
/* aes_hash.rs:59-66: 193502124791825095790518994062991136444 */
static const uint8_t KEY_HASH[16] = {0xbc, 0x96, 0x17, 0x9a, 0x4d, 0x75, 0xfc, 0xd2,
                                     0x15, 0x49, 0x1e, 0x35, 0x55, 0x27, 0x93, 0x91};

/* aes_hash.rs:47-55 */
void orc_cr_hash(orc_block *x, size_t n) {
    aes_key k;
    aes_init(&k, KEY_HASH);
    const size_t CH = 4096;
    orc_block *tmp = (orc_block *)malloc(CH * sizeof(orc_block));
    for (size_t i = 0; i < n; i += CH) {
        size_t m = n - i < CH ? n - i : CH;
        aes_encrypt_n(&k, (const uint8_t *)(x + i), (uint8_t *)tmp, m);
        for (size_t j = 0; j < m; j++) x[i + j] = bxor(x[i + j], tmp[j]);
    }
    free(tmp);
}

/* ------------------------------------------------------------------ */
/* configuration (quasi_cyclic_encode.rs:177-193, mod.rs:671-695)      */
/* ------------------------------------------------------------------ */
static int is_prime(uint64_t x) {
    if (x < 2) return 0;
    if (x % 2 == 0) return x == 2;
    for (uint64_t d = 3; d * d <= x; d += 2)
        if (x % d == 0) return 0;
    return 1;
}
/* num_prime::nt_funcs::next_prime 0.4.3: smallest prime strictly greater than x */
uint64_t orc_next_prime(uint64_t x) {
    uint64_t c = x + 1;
    while (!is_prime(c)) c++;
    return c;
}
/* mod.rs:690-695, same double operations */
static uint64_t sec_level(uint64_t scale, uint64_t p, uint64_t points) {
    double x1 = log2((double)(scale * p) / (double)p);
    double x2 = log2((double)(scale * p)) / 2.0;
    double s = (double)points * x1 + x2;
    return (uint64_t)s;
}
/* mod.rs:671-681 */
static uint64_t get_partitions(uint64_t scaler, uint64_t prime, uint64_t sec_param) {
    uint64_t ret = 1;
    uint64_t ss = sec_level(scaler, prime, ret);
    while (ss < sec_param) {
        ret += 1;
        ss = sec_level(scaler, prime, ret);
        if (ret > 1000) return 0;
    }
    return (ret + 7) / 8 * 8;
}
static uint64_t log2_ceil(uint64_t x) {
    uint64_t d = 0;
    while ((1ULL << d) < x) d++;
    return d;
}
int orc_configure(uint64_t n, uint64_t scaler, uint64_t sec_param, orc_conf *c) {
    if (scaler < 2 || n == 0) return -1;
    memset(c, 0, sizeof(*c));
    c->n = n;
    c->scaler = scaler;
    uint64_t m = n > 128 * 128 ? n : 128 * 128;
    c->P = orc_next_prime(m);
    c->parts = get_partitions(scaler, c->P, sec_param);
    if (c->parts == 0) return -2;
    uint64_t ss = (c->P * scaler + c->parts - 1) / c->parts;
    c->size_per = (ss + 7) / 8 * 8;
    c->N2 = c->size_per * c->parts;
    c->N = c->N2 / scaler;
    c->depth = log2_ceil(c->size_per);
    c->n_blocks = c->N / 128;
    c->n64 = 2 * c->n_blocks;
    c->cols = (c->parts * c->size_per + 127) / 128;
    c->base_ot_count = c->depth * c->parts;
    return 0;
}

/* ------------------------------------------------------------------ */
/* transpose                                                           */
/* ------------------------------------------------------------------ */
/* util/transpose.rs:63-64,93-100: out[(c)*nrows/8 + r/8] bit r%8 = in[r*ncols/8 + c/8] bit c%8 */
void orc_transpose_naive(const uint8_t *in, uint8_t *out, uint64_t nrows, uint64_t ncols) {
    memset(out, 0, nrows * ncols / 8);
    for (uint64_t r = 0; r < nrows; r++)
        for (uint64_t c = 0; c < ncols; c++) {
            uint8_t bit = (in[r * ncols / 8 + c / 8] >> (c % 8)) & 1;
            out[c * nrows / 8 + r / 8] |= (uint8_t)(bit << (r % 8));
        }
}

/* 8x8 bit transpose, byte i = row i, bit j = column j (LSB first) */
static inline uint64_t tr8x8(uint64_t x) {
    uint64_t t;
    t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAULL;
    x = x ^ t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCULL;
    x = x ^ t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ULL;
    x = x ^ t ^ (t << 28);
    return x;
}
static void transpose_128x128(const uint8_t *in, uint8_t *out) {
    for (int rb = 0; rb < 16; rb++)
        for (int cb = 0; cb < 16; cb++) {
            uint64_t x = 0;
            for (int i = 0; i < 8; i++) x |= (uint64_t)in[(rb * 8 + i) * 16 + cb] << (8 * i);
            x = tr8x8(x);
            for (int j = 0; j < 8; j++) out[(cb * 8 + j) * 16 + rb] = (uint8_t)(x >> (8 * j));
        }
}
void orc_transpose_128(orc_block m[128]) {
    orc_block tmp[128];
    transpose_128x128((const uint8_t *)m, (uint8_t *)tmp);
    memcpy(m, tmp, sizeof(tmp));
}

/* ------------------------------------------------------------------ */
/* PPRF                                                                */
/* ------------------------------------------------------------------ */
uint64_t orc_get_active_path(const uint8_t *cb, uint64_t depth) {
    uint64_t point = 0;
    for (uint64_t i = 0; i < depth; i++) point |= (uint64_t)(1 ^ cb[i]) << (depth - i - 1);
    return point;
}
uint64_t orc_interleave_point(uint64_t point, uint64_t tree_idx, uint64_t total_trees) {
    uint64_t num_sets = total_trees / 8;
    uint64_t set_idx = tree_idx / 8, sub_idx = tree_idx % 8;
    uint64_t section_idx = point / 16, pos_idx = point % 16;
    return set_idx * 128 + (sub_idx + 8 * pos_idx) + section_idx * num_sets * 128;
}
void orc_pprf_points(const orc_conf *c, const uint8_t *choices, uint64_t *points_out) {
    for (uint64_t t = 0; t < c->parts; t++) {
        uint64_t p = orc_get_active_path(choices + t * c->depth, c->depth);
        points_out[t] = orc_interleave_point(p, t, c->parts);
    }
}

/* pprf.rs:790-835 (InterleavedTransposed). lvl: leaves [node][8], n_leaves nodes */
static void pprf_copy_out(const orc_block *lvl, uint64_t n_leaves, orc_block *out, uint64_t cols, uint64_t total_trees,
                          uint64_t t_idx) {
    uint64_t set_idx = t_idx / 8;
    uint64_t block_per_set = n_leaves * 8 / 128;
    uint64_t step = total_trees / 8;
    uint64_t end = set_idx + step * block_per_set;
    if (end > cols) end = cols;
    uint64_t i = set_idx, k = 0;
    orc_block tile[128];
    while (i < end) {
        transpose_128x128((const uint8_t *)(lvl + k * 16 * 8), (uint8_t *)tile);
        for (int r = 0; r < 128; r++) out[(uint64_t)r * cols + i] = tile[r];
        i += step;
        k += 1;
    }
}

/* one GGM level for 8 interleaved trees: pprf.rs:154-193 */
static void ggm_level(const aes_key aes[2], const orc_block *level0, orc_block *level1, uint64_t width0,
                      orc_block sum0[8], orc_block sum1[8]) {
    for (uint64_t j = 0; j < width0; j++) {
        const orc_block *parent = level0 + 8 * j;
        for (int keep = 0; keep < 2; keep++) {
            orc_block *child = level1 + 8 * (2 * j + keep);
            aes_encrypt_n(&aes[keep], (const uint8_t *)parent, (uint8_t *)child, 8);
            orc_block *sum = keep ? sum1 : sum0;
            for (int t = 0; t < 8; t++) {
                child[t] = bxor(child[t], parent[t]);
                sum[t] = bxor(sum[t], child[t]);
            }
        }
    }
}

int orc_pprf_sender_expand(const orc_conf *c, const orc_block *base_ots, orc_block delta, orc_block seed,
                           uint64_t num_threads, orc_block *out_rT, orc_block *tg_sums, orc_block *tg_last) {
    const uint64_t depth = c->depth, pnt = c->parts, cols = c->cols;
    if (pnt % 8 || depth < 2 || num_threads == 0) return -1;
    aes_key aes[2], kh;
    aes_init(&aes[0], KEY_GGM0);
    aes_init(&aes[1], KEY_GGM1);
    aes_init(&kh, KEY_HASH);
    memset(out_rT, 0, 128 * cols * sizeof(orc_block)); /* pprf.rs:93 */
    int failed = 0;
    /* pprf.rs:267-276: (0..num_threads).into_par_iter().for_each(routine) on the instance's rayon pool: the logical
     * workers run on as many OS threads as orc_set_num_threads() gave this caller (1 by default); every worker owns its
     * tree scratch (pprf.rs:116-126) and writes disjoint columns of the output (the reference takes a Mutex, :261) */
#pragma omp parallel for schedule(dynamic, 1)
    for (uint64_t t = 0; t < num_threads; t++) {
        orc_block *tree = (orc_block *)malloc(((size_t)8 << (depth + 1)) * sizeof(orc_block));
        orc_block(*sums)[8] = (orc_block(*)[8])malloc(2 * depth * 8 * sizeof(orc_block));
        if (!tree || !sums) {
            failed = 1;
            free(tree);
            free(sums);
            continue;
        }
        /* pprf.rs:115: AesRng::from_seed(seed ^ thread_idx) */
        orc_block s = seed;
        s.w[0] ^= t;
        uint64_t draws = 0;
        for (uint64_t g = t * 8; g < pnt; g += 8 * num_threads, draws++) {
            /* pprf.rs:135-137: level 0 <- 128 bytes of the stream (blocks 8*draws ..) */
            orc_aes_rng_blocks((const uint8_t *)&s, 8 * draws, (uint8_t *)tree, 8);
            memset(sums, 0, 2 * depth * 8 * sizeof(orc_block));
            for (uint64_t d = 0; d < depth; d++) {
                orc_block *l0 = tree + 8 * (((uint64_t)1 << d) - 1);
                orc_block *l1 = tree + 8 * (((uint64_t)1 << (d + 1)) - 1);
                ggm_level(aes, l0, l1, (uint64_t)1 << d, sums[0 * depth + d], sums[1 * depth + d]);
            }
            uint64_t grp = g / 8;
            /* pprf.rs:199-212: mask sums of levels 0..depth-2 with base OTs */
            for (int keep = 0; keep < 2; keep++)
                for (uint64_t d = 0; d + 1 < depth; d++)
                    for (int j = 0; j < 8; j++) {
                        orc_block v = bxor(sums[keep * depth + d][j], base_ots[((g + j) * depth + d) * 2 + keep]);
                        tg_sums[((grp * 2 + keep) * (depth - 1) + d) * 8 + j] = v;
                    }
            /* pprf.rs:222-250 */
            uint64_t d = depth - 1;
            for (int j = 0; j < 8; j++) {
                orc_block s0 = sums[0 * depth + d][j], s1 = sums[1 * depth + d][j];
                orc_block lo[4] = {s0, bxor(s1, delta), s1, bxor(s0, delta)};
                orc_block o0 = base_ots[((g + j) * depth + d) * 2 + 0];
                orc_block o1 = base_ots[((g + j) * depth + d) * 2 + 1];
                orc_block mask_in[4] = {o0, bxor(o0, bones_()), o1, bxor(o1, bones_())};
                orc_block enc[4];
                aes_encrypt_n(&kh, (const uint8_t *)mask_in, (uint8_t *)enc, 4);
                for (int q = 0; q < 4; q++)
                    tg_last[(grp * 8 + j) * 4 + q] = bxor(lo[q], bxor(enc[q], mask_in[q]));
            }
            /* pprf.rs:259-263 */
            pprf_copy_out(tree + 8 * (((uint64_t)1 << depth) - 1), (uint64_t)1 << depth, out_rT, cols, pnt, g);
        }
        free(tree);
        free(sums);
    }
    return failed ? -2 : 0;
}

int orc_pprf_receiver_expand(const orc_conf *c, const orc_block *base_ots, const uint8_t *choices,
                             const orc_block *tg_sums, const orc_block *tg_last, orc_block *out_rT) {
    const uint64_t depth = c->depth, pnt = c->parts, cols = c->cols;
    if (pnt % 8 || depth < 2) return -1;
    aes_key aes[2], kh;
    aes_init(&aes[0], KEY_GGM0);
    aes_init(&aes[1], KEY_GGM1);
    aes_init(&kh, KEY_HASH);
    memset(out_rT, 0, 128 * cols * sizeof(orc_block));
    /* pprf.rs:315: plain points */
    uint64_t *points = (uint64_t *)malloc(pnt * sizeof(uint64_t));
    for (uint64_t t = 0; t < pnt; t++) points[t] = orc_get_active_path(choices + t * depth, depth);
    int failed = 0;
    /* pprf.rs:347-372,546-555: the receiver's workers take tree groups as their messages arrive; here every group is a
     * task on the caller's OpenMP team (the result does not depend on the worker count) */
#pragma omp parallel
    {
    orc_block *tree = (orc_block *)calloc((size_t)8 << (depth + 1), sizeof(orc_block));
    if (!tree) failed = 1;
#pragma omp for schedule(dynamic, 1)
    for (uint64_t grp_i = 0; grp_i < pnt / 8; grp_i++) {
        if (!tree) continue;
        const uint64_t g = 8 * grp_i;
        uint64_t grp = g / 8;
        const orc_block *gs = tg_sums + grp * 2 * (depth - 1) * 8; /* [keep][d][8] */
        orc_block my_sums[2][8];
        /* pprf.rs:376-384 */
        orc_block *l1 = tree + 8 * 1;
        for (int i = 0; i < 8; i++) {
            unsigned not_ai = choices[(g + i) * depth + 0];
            l1[8 * not_ai + i] = bxor(base_ots[(g + i) * depth + 0], gs[(not_ai * (depth - 1) + 0) * 8 + i]);
            l1[8 * (not_ai ^ 1) + i] = bzero_();
        }
        memset(my_sums, 0, sizeof(my_sums));
        /* pprf.rs:388-468 */
        for (uint64_t d = 1; d < depth; d++) {
            orc_block *l0 = tree + 8 * (((uint64_t)1 << d) - 1);
            orc_block *lv1 = tree + 8 * (((uint64_t)1 << (d + 1)) - 1);
            memset(my_sums, 0, sizeof(my_sums));
            ggm_level(aes, l0, lv1, (uint64_t)1 << d, my_sums[0], my_sums[1]);
            if (d != depth - 1) {
                for (int i = 0; i < 8; i++) {
                    uint64_t leaf_idx = points[g + i];
                    uint64_t active_child_idx = leaf_idx >> (depth - 1 - d);
                    uint64_t inactive_child_idx = active_child_idx ^ 1;
                    unsigned not_ai = inactive_child_idx & 1;
                    orc_block *inactive = &lv1[8 * inactive_child_idx + i];
                    orc_block correct_sum = bxor(*inactive, gs[(not_ai * (depth - 1) + d) * 8 + i]);
                    *inactive = bxor(bxor(correct_sum, my_sums[not_ai][i]), base_ots[(g + i) * depth + d]);
                }
            }
        }
        /* pprf.rs:474-535 */
        orc_block *level = tree + 8 * (((uint64_t)1 << depth) - 1);
        uint64_t d = depth - 1;
        for (int j = 0; j < 8; j++) {
            uint64_t active_child_idx = points[g + j];
            uint64_t inactive_child_idx = active_child_idx ^ 1;
            unsigned not_ai = inactive_child_idx & 1;
            orc_block o = base_ots[(g + j) * depth + d];
            orc_block mask_in[2] = {o, bxor(o, bones_())};
            orc_block enc[2];
            aes_encrypt_n(&kh, (const uint8_t *)mask_in, (uint8_t *)enc, 2);
            orc_block ots[2];
            for (int q = 0; q < 2; q++)
                ots[q] = bxor(tg_last[(grp * 8 + j) * 4 + 2 * not_ai + q], bxor(enc[q], mask_in[q]));
            orc_block *inactive_child = &level[8 * inactive_child_idx + j];
            orc_block *active_child = &level[8 * active_child_idx + j];
            orc_block inactive_sum = bxor(my_sums[not_ai][j], *inactive_child);
            orc_block active_sum = bxor(my_sums[not_ai ^ 1][j], *active_child);
            *inactive_child = bxor(ots[0], inactive_sum);
            *active_child = bxor(ots[1], active_sum);
        }
        pprf_copy_out(level, (uint64_t)1 << depth, out_rT, cols, pnt, g);
    }
    free(tree);
    }
    free(points);
    return failed ? -2 : 0;
}

/* ------------------------------------------------------------------ */
/* GF(2)[x] product                                                    */
/* ------------------------------------------------------------------ */
void orc_gf2x_mul_naive(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n) {
    memset(c, 0, 2 * n * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++)
        for (int bit = 0; bit < 64; bit++)
            if ((a[i] >> bit) & 1) {
                for (uint64_t j = 0; j < n; j++) {
                    c[i + j] ^= b[j] << bit;
                    if (bit) c[i + j + 1] ^= b[j] >> (64 - bit);
                }
            }
}

/* 64x64 -> 128 carry-less, portable */
static inline void clmul64_portable(uint64_t a, uint64_t b, uint64_t *lo, uint64_t *hi) {
    uint64_t l = 0, h = 0;
    for (int i = 0; i < 64; i++)
        if ((a >> i) & 1) {
            l ^= b << i;
            if (i) h ^= b >> (64 - i);
        }
    *lo = l;
    *hi = h;
}

static void school_portable(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n) {
    memset(c, 0, 2 * n * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++)
        for (uint64_t j = 0; j < n; j++) {
            uint64_t lo, hi;
            clmul64_portable(a[i], b[j], &lo, &hi);
            c[i + j] ^= lo;
            c[i + j + 1] ^= hi;
        }
}

#if ORC_X86
__attribute__((target("pclmul,sse4.1"))) static void school_clmul(uint64_t *c, const uint64_t *a, const uint64_t *b,
                                                                  uint64_t n) {
    memset(c, 0, 2 * n * sizeof(uint64_t));
    for (uint64_t i = 0; i < n; i++) {
        __m128i ai = _mm_set_epi64x(0, (long long)a[i]);
        for (uint64_t j = 0; j < n; j++) {
            __m128i bj = _mm_set_epi64x(0, (long long)b[j]);
            __m128i p = _mm_clmulepi64_si128(ai, bj, 0);
            __m128i cur = _mm_loadu_si128((const __m128i *)(c + i + j));
            _mm_storeu_si128((__m128i *)(c + i + j), _mm_xor_si128(cur, p));
        }
    }
}
#endif

static void school(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n) {
#if ORC_X86
    if (have_pclmul()) {
        school_clmul(c, a, b, n);
        return;
    }
#endif
    school_portable(c, a, b, n);
}

/* balanced Karatsuba, n = m * 2^k; scratch needs 4n words */
static void kara(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t base, uint64_t *scratch) {
    if (n <= base || (n & 1)) {
        school(c, a, b, n);
        return;
    }
    uint64_t h = n / 2;
    uint64_t *sa = scratch, *sb = scratch + h, *mid = scratch + 2 * h, *next = scratch + 4 * h;
    kara(c, a, b, h, base, next);                 /* z0 */
    kara(c + 2 * h, a + h, b + h, h, base, next); /* z2 */
    for (uint64_t i = 0; i < h; i++) {
        sa[i] = a[i] ^ a[i + h];
        sb[i] = b[i] ^ b[i + h];
    }
    kara(mid, sa, sb, h, base, next);
    for (uint64_t i = 0; i < 2 * h; i++) mid[i] ^= c[i] ^ c[2 * h + i];
    for (uint64_t i = 0; i < 2 * h; i++) c[h + i] ^= mid[i];
}

static orc_mul_fn g_ext_mul = NULL;
void orc_set_external_mul(orc_mul_fn fn) { g_ext_mul = fn; }

void orc_gf2x_mul(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n64) {
    if (g_ext_mul && n64 >= 256) { /* bitpolymul2 aborts below 256 words: encode.c:102 */
        /* bitpolymul needs 32-byte aligned operands (bitpolymul/src/lib.rs:13,47) */
        uint64_t *aa, *bb, *cc;
        if (posix_memalign((void **)&aa, 32, n64 * 8) || posix_memalign((void **)&bb, 32, n64 * 8) ||
            posix_memalign((void **)&cc, 32, 2 * n64 * 8))
            abort();
        memcpy(aa, a, n64 * 8);
        memcpy(bb, b, n64 * 8);
        g_ext_mul(cc, aa, bb, (unsigned)n64);
        memcpy(c, cc, 2 * n64 * 8);
        free(aa);
        free(bb);
        free(cc);
        return;
    }
    /* pad to m * 2^k with m <= 32 */
    uint64_t k = 0;
    while (((n64 + ((uint64_t)1 << k) - 1) >> k) > 32) k++;
    uint64_t np = ((n64 + ((uint64_t)1 << k) - 1) >> k) << k;
    uint64_t *buf = (uint64_t *)calloc(2 * np + 2 * np + 8 * np + 64, sizeof(uint64_t));
    if (!buf) abort();
    uint64_t *pa = buf, *pb = buf + np, *pc = buf + 2 * np, *scr = buf + 4 * np;
    memcpy(pa, a, n64 * 8);
    memcpy(pb, b, n64 * 8);
    kara(pc, pa, pb, np, np >> k, scr);
    memcpy(c, pc, 2 * n64 * 8);
    free(buf);
}

/* ------------------------------------------------------------------ */
/* QuasiCyclic encode                                                  */
/* ------------------------------------------------------------------ */
/* quasi_cyclic_encode.rs:329-344 */
void orc_bit_shift_xor(orc_block *dest, uint64_t dest_len, const orc_block *inp, uint64_t inp_len, uint8_t bit_shift) {
    /* zip(dest, inp, inp[1..]) */
    uint64_t m = dest_len;
    if (inp_len < m) m = inp_len;
    if (inp_len - 1 < m) m = inp_len - 1;
    for (uint64_t i = 0; i < m; i++) {
        orc_block shifted = bshr(inp[i], bit_shift);
        orc_block up = bshl(inp[i + 1], 128 - bit_shift);
        shifted.w[0] |= up.w[0];
        shifted.w[1] |= up.w[1];
        dest[i] = bxor(dest[i], shifted);
    }
    if (dest_len >= inp_len) {
        orc_block last = inp[inp_len - 1];
        dest[inp_len - 1] = bxor(dest[inp_len - 1], bshr(last, bit_shift));
    }
}

/* quasi_cyclic_encode.rs:289-327 */
void orc_modp(orc_block *dest, uint64_t dest_len, const orc_block *inp, uint64_t inp_len, uint64_t p) {
    uint64_t p_bytes = (p + 7) / 8;
    uint64_t count = (inp_len * 128 + p - 1) / p;
    memcpy(dest, inp, p_bytes);
    for (uint64_t i = 1; i < count; i++) {
        uint64_t begin = i * p;
        uint64_t begin_block = begin / 128;
        uint64_t end_bit = i * p + p;
        if (end_bit > inp_len * 128) end_bit = inp_len * 128;
        uint64_t end_block = (end_bit + 127) / 128;
        unsigned shift = begin & 127;
        orc_bit_shift_xor(dest, dest_len, inp + begin_block, end_block - begin_block, (uint8_t)shift);
    }
    uint8_t *db = (uint8_t *)dest;
    unsigned offset = p & 7;
    if (offset) db[p / 8] &= (uint8_t)((1u << offset) - 1);
    uint64_t rem = dest_len * 16 - p_bytes;
    if (rem) memset(db + p_bytes, 0, rem);
}

/* quasi_cyclic_encode.rs:127-138: a_s = the first n64 u64 of AesRng::from_seed(Block::from(s)), s = 1 .. scaler - 1 */
void orc_qc_a_poly_s(const orc_conf *c, uint64_t sidx, uint64_t *a_out) {
    uint8_t seed[16];
    u128_key(sidx, 0, seed);
    uint64_t nblk = (c->n64 + 1) / 2;
    uint8_t *tmp = (uint8_t *)malloc(nblk * 16);
    orc_aes_rng_blocks(seed, 0, tmp, nblk);
    memcpy(a_out, tmp, c->n64 * 8);
    free(tmp);
}
void orc_qc_a_poly(const orc_conf *c, uint64_t *a_out) { orc_qc_a_poly_s(c, 1, a_out); }

/* per calling thread: two host threads run the two directions concurrently (bench.py cpu_baseline) */
static __thread orc_block *g_cmodp = NULL;
static __thread uint64_t g_cmodp_rows = 0, g_cmodp_cols = 0;
const orc_block *orc_last_cmodp(uint64_t *rows, uint64_t *cols) {
    *rows = g_cmodp_rows;
    *cols = g_cmodp_cols;
    return g_cmodp;
}

/* quasi_cyclic_encode.rs:262-286. a: the scaler - 1 public polynomials, n64 words each; the reference sums the products in
 * the transform domain (c_poly.add_eq), which equals the XOR of the products */
static void qc_reduce(const orc_conf *c, const uint64_t *a, orc_block *dest, const orc_block *b128) {
    uint64_t n64 = c->n64, nb = c->n_blocks;
    orc_block *temp = (orc_block *)calloc(2 * nb, sizeof(orc_block));
    orc_block *prod = c->scaler > 2 ? (orc_block *)malloc(2 * nb * sizeof(orc_block)) : NULL;
    for (uint64_t s = 1; s < c->scaler; s++) {
        const uint64_t *b64 = (const uint64_t *)b128 + s * n64; /* [s*n64 .. (s+1)*n64) */
        if (s == 1) {
            orc_gf2x_mul((uint64_t *)temp, a, b64, n64);
        } else {
            orc_gf2x_mul((uint64_t *)prod, a + (s - 1) * n64, b64, n64);
            for (uint64_t i = 0; i < 2 * nb; i++) temp[i] = bxor(temp[i], prod[i]);
        }
    }
    free(prod);
    for (uint64_t i = 0; i < nb; i++) temp[i] = bxor(temp[i], b128[i]);
    orc_modp(dest, nb, temp, 2 * nb, c->P);
    free(temp);
}

/* quasi_cyclic_encode.rs:140-151 */
static void qc_copy_out(orc_block *dest, uint64_t num_blocks, const orc_block *cmodp, uint64_t nb) {
#pragma omp parallel for schedule(static)
    for (uint64_t i = 0; i < num_blocks / 128; i++) {
        orc_block *chunk = dest + 128 * i;
        for (int r = 0; r < 128; r++) chunk[r] = cmodp[(uint64_t)r * nb + i];
        orc_transpose_128(chunk);
    }
}

static int qc_encode_common(const orc_conf *c, const orc_block *rT, const orc_block *sb, int packed, orc_block *out,
                            uint8_t *out_C) {
    if (c->scaler < 2) return -1;
    /* the reference's modp copies ceil(P / 8) bytes into a dest of n_blocks Blocks (quasi_cyclic_encode.rs:289-303): a geometry
     * with n_blocks * 128 < P panics there; refuse it here */
    if (c->n_blocks * 128 < c->P) return -2;
    uint64_t nb = c->n_blocks;
    uint64_t *a = (uint64_t *)malloc((c->scaler - 1) * c->n64 * 8);
    for (uint64_t s = 1; s < c->scaler; s++) orc_qc_a_poly_s(c, s, a + (s - 1) * c->n64);
    free(g_cmodp);
    g_cmodp = (orc_block *)calloc(128 * nb, sizeof(orc_block));
    g_cmodp_rows = 128;
    g_cmodp_cols = nb;
    orc_block *const cmodp = g_cmodp; /* the parallel regions below read a plain pointer, not the thread-local */
#pragma omp parallel for schedule(dynamic)
    for (int r = 0; r < 128; r++) {
        const orc_block *row = (sb && packed && r == 0) ? sb : rT + (uint64_t)r * c->cols;
        qc_reduce(c, a, cmodp + (uint64_t)r * nb, row);
    }
    if (sb && !packed) {
        /* quasi_cyclic_encode.rs:104-117 */
        orc_block *c128 = (orc_block *)calloc(nb, sizeof(orc_block));
        qc_reduce(c, a, c128, sb);
        const uint8_t *bytes = (const uint8_t *)c128;
        for (uint64_t i = 0; i < c->n; i++) out_C[i] = (bytes[i / 8] >> (i % 8)) & 1;
        free(c128);
    }
    uint64_t num_blocks = (c->n + 127) / 128 * 128;
    orc_block *full = (orc_block *)calloc(c->N2 > num_blocks ? c->N2 : num_blocks, sizeof(orc_block));
    qc_copy_out(full, num_blocks, cmodp, nb);
    memcpy(out, full, c->n * sizeof(orc_block));
    free(full);
    free(a);
    return 0;
}

int orc_qc_dual_encode(const orc_conf *c, const orc_block *rT, orc_block *out_B) {
    return qc_encode_common(c, rT, NULL, 0, out_B, NULL);
}

int orc_qc_dual_encode2(const orc_conf *c, const orc_block *rT, const uint64_t *S, uint64_t nS, int packed,
                        orc_block *out_A, uint8_t *out_C) {
    /* mod.rs:560-574: N2 BYTES allocated (size_of::<u8>() divisor), bits set Lsb0 */
    uint8_t *sb = (uint8_t *)calloc(c->N2 + 16, 1);
    for (uint64_t i = 0; i < nS; i++) sb[S[i] / 8] |= (uint8_t)(1u << (S[i] % 8));
    int rc = qc_encode_common(c, rT, (const orc_block *)sb, packed, out_A, out_C);
    free(sb);
    return rc;
}

/* ------------------------------------------------------------------ */
/* hashes + MT bits                                                    */
/* ------------------------------------------------------------------ */
void orc_sender_hash(orc_block delta, const orc_block *B, uint64_t n, orc_block *out) {
    orc_block d = delta;
    d.w[0] &= ~1ULL;
    for (uint64_t i = 0; i < n; i++) {
        orc_block m = B[i];
        m.w[0] &= ~1ULL;
        out[2 * i] = m;
        out[2 * i + 1] = bxor(m, d);
    }
    orc_cr_hash(out, 2 * n);
}

void orc_receiver_hash(orc_block *A, uint64_t n, uint64_t *choice_words) {
    memset(choice_words, 0, (n + 63) / 64 * 8);
    for (uint64_t i = 0; i < n; i++) {
        choice_words[i / 64] |= (A[i].w[0] & 1ULL) << (i % 64);
        A[i].w[0] &= ~1ULL;
    }
    orc_cr_hash(A, n);
}

void orc_mt_bits(const orc_block *send_msgs, const orc_block *recv_msgs, const uint64_t *recv_choice_words, uint64_t n,
                 uint64_t *a, uint64_t *b, uint64_t *c) {
    uint64_t nw = (n + 63) / 64;
    memset(a, 0, nw * 8);
    memset(b, 0, nw * 8);
    memset(c, 0, nw * 8);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t m0 = send_msgs[2 * i].w[0] & 1, m1 = send_msgs[2 * i + 1].w[0] & 1;
        uint64_t ai = (recv_choice_words[i / 64] >> (i % 64)) & 1;
        uint64_t bi = m0 ^ m1, vi = m0, ui = recv_msgs[i].w[0] & 1;
        uint64_t ci = (ai & bi) ^ ui ^ vi;
        a[i / 64] |= ai << (i % 64);
        b[i / 64] |= bi << (i % 64);
        c[i / 64] |= ci << (i % 64);
    }
}
