// QuasiCyclic dual-LPN encode on gfx950: B[k] = V[k] ^ V[k+P] ^ ((a * V1) mod (x^P - 1))[k]
//
// Reference: zappot/src/silent_ot/quasi_cyclic_encode.rs:29-151,262-344 (128 per-row products
// a * b1_r via bitpolymul's additive FFT over GF(2^128), + b0_r, modp, 128x128 transposes).
//
// MI355X formulation: the 128 rows multiply by the same public a, and a Block of the untransposed
// PPRF output V holds one coefficient of all 128 row polynomials, so the rows are processed bit-sliced
// (lane = row, bit = 32 consecutive FFT points) and every field operation is an XOR/AND network:
//   V1 -> 16 bit planes (Kronecker x -> g, 16 bits per element) -> monomial-to-novel basis conversion
//   -> GF(2^32) (nimber) coordinates, 32 planes -> LCH additive FFT butterflies with Cantor-basis points
//   -> pointwise multiply by FFT(a) (precomputed bit matrices) -> inverse butterflies -> inverse basis
//   conversion -> overlap-add back to Blocks T -> fold mod x^P - 1 together with V0 and the short
//   schoolbook "tail" terms for the few bits of a and V1 above the power-of-two FFT capacity.
// The math (and the generators of the constant tables) is in gf2fft_model.py / gen_tables.py.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sot_internal.h"

#define ROWS 128  // bit rows of a Block; lane l of a wave owns rows l and l + 64

static __device__ const uint32_t d_BETA[32] = {
#include "qc_beta.inc"
};
static __device__ const uint32_t d_RTAB[32 * 32] = {
#include "qc_rtab.inc"
};
static __device__ const uint32_t d_MPOS[4 * 8 * 8] = {
#include "qc_mpos.inc"
};

// ------------------------------------------------------------------------------------------------
// helpers
// a ^ (b & c) in one VALU op (gfx950 v_bitop3_b32, truth table 0xF0 ^ (0xCC & 0xAA))
#define XAND(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x78)
// 32x32 bit transpose, LSB first: on return a[i] bit p == (old a[p]) bit i
__device__ __forceinline__ void transpose32(uint32_t a[32]) {
    uint32_t m = 0x0000FFFFu;
#pragma unroll
    for (int j = 16; j != 0; j >>= 1, m ^= (m << j)) {
#pragma unroll
        for (int k = 0; k < 32; k = (k + j + 1) & ~j) {
            uint32_t t = ((a[k] >> j) ^ a[k + j]) & m;
            a[k] ^= t << j;
            a[k + j] ^= t;
        }
    }
}

// bits b in [0,32) with lo <= b < hi (arguments may be out of range / negative)
__device__ __forceinline__ uint32_t range_mask(int64_t lo, int64_t hi) {
    if (lo < 0) lo = 0;
    if (hi > 32) hi = 32;
    if (hi <= lo) return 0u;
    uint32_t top = hi == 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u);
    return top & ~((1u << lo) - 1u);  // lo < 32 here
}

// ------------------------------------------------------------------------------------------------
// receiver with packed choice bits: bit 0 of the leaf vector is replaced by the noise indicator sb
// (silent_ot/mod.rs:560-574, quasi_cyclic_encode.rs:92-100)
__global__ void k_clear_lsb(blk_t *__restrict__ V, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t *p = (uint32_t *)(V + i);
        *p &= ~1u;
    }
}
__global__ void k_set_points(blk_t *__restrict__ V, const uint64_t *__restrict__ pts, uint32_t npts, uint64_t nV) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npts && pts[i] < nV) {
        uint32_t *p = (uint32_t *)(V + pts[i]);
        atomicOr(p, 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// K1: Blocks -> 16 bit planes. X[(w*16 + t)*ROWS + row] bit p = bit `row` of V1'[16*(32 w + p) + t],
// V1'[c] = V1[c] ^ V1[c + P] (reduction mod x^P - 1 of the 2*n_blocks*128-bit row), c < lo_bits.
// One wave per word index w (512 Blocks in, 8 KiB out).
__global__ void __launch_bounds__(256)
    k_qc_load(const blk_t *__restrict__ V1, uint64_t v1_len, uint64_t P, uint64_t lo_bits, uint64_t nw_in,
              uint32_t *__restrict__ X) {
    __shared__ uint32_t tile[4][512 * 4];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t w = (uint64_t)blockIdx.x * 4 + wave;
    if (w < nw_in) {
        for (uint32_t i = lane; i < 512; i += 64) {
            uint64_t c = w * 512 + i;
            blk_t v = make_uint4(0, 0, 0, 0);
            if (c < lo_bits) {
                v = V1[c];
                if (c + P < v1_len) {
                    blk_t u = V1[c + P];
                    v.x ^= u.x;
                    v.y ^= u.y;
                    v.z ^= u.z;
                    v.w ^= u.w;
                }
            }
            *(blk_t *)&tile[wave][i * 4] = v;
        }
    }
    __syncthreads();
    if (w >= nw_in) return;
    const uint32_t t = lane >> 2, q = lane & 3;
    uint32_t a[32];
#pragma unroll
    for (int p = 0; p < 32; p++) a[p] = tile[wave][(16 * p + t) * 4 + q];
    transpose32(a);
    uint32_t *dst = X + (w * 16 + t) * ROWS + 32 * q;
#pragma unroll
    for (int b = 0; b < 32; b++) dst[b] = a[b];
}

// public polynomial: row 0 of the plane layout from a packed bit vector, other rows zero (memset before)
__global__ void k_a_load(const uint32_t *__restrict__ abits, uint64_t nbits, uint64_t nw_in, uint32_t *__restrict__ X) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (w, t)
    if (idx >= nw_in * 16) return;
    uint64_t w = idx >> 4;
    uint32_t t = idx & 15;
    uint32_t word = 0;
    for (uint32_t p = 0; p < 32; p++) {
        uint64_t bit = 16 * (32 * w + p) + t;
        if (bit < nbits) word |= ((abits[bit >> 5] >> (bit & 31)) & 1u) << p;
    }
    X[(w * 16 + t) * ROWS] = word;
}

// ------------------------------------------------------------------------------------------------
// K2: one phase of the monomial <-> novel basis conversion (gf2fft_model.conversion_phases):
//   out[i] = in[i] ^ m1(l) in[i + D] ^ m2(l) in[i + 2D],  l = i mod chunk, D = half - pw
//   m1 = (l >= half) ? (l - half < pw) : (l >= pw);  m2 = forward && pw <= l < 2 pw
// i indexes elements = bit positions along a plane. Thread = (word, plane, row).
template <int NP>
__global__ void __launch_bounds__(256)
    k_conv_phase(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint64_t nwords, uint64_t chunk,
                 uint64_t pw, uint64_t half, int inverse) {
    const uint64_t total = nwords * NP * ROWS;
    const uint64_t D = half - pw;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t wd = idx / (NP * ROWS);
        const uint64_t rem = idx - wd * (NP * ROWS);  // plane*ROWS + row
        uint32_t m1, m2;
        if (chunk >= 32) {
            int64_t l0 = (int64_t)((32 * wd) & (chunk - 1));
            m1 = range_mask((int64_t)half - l0, (int64_t)(half + pw) - l0) | range_mask((int64_t)pw - l0, (int64_t)half - l0);
            m2 = inverse ? 0u : range_mask((int64_t)pw - l0, (int64_t)(2 * pw) - l0);
        } else {
            uint32_t c1 = range_mask((int64_t)half, (int64_t)(half + pw)) | range_mask((int64_t)pw, (int64_t)half);
            uint32_t c2 = inverse ? 0u : range_mask((int64_t)pw, (int64_t)(2 * pw));
            for (uint32_t s = (uint32_t)chunk; s < 32; s <<= 1) {
                c1 |= c1 << s;
                c2 |= c2 << s;
            }
            m1 = c1;
            m2 = c2;
        }
        uint32_t v = in[idx];
        if (m1) {
            uint64_t ws = wd + (D >> 5);
            uint32_t sh = D & 31;
            uint32_t lo = ws < nwords ? in[ws * (NP * ROWS) + rem] : 0u;
            uint32_t hi = (sh && ws + 1 < nwords) ? in[(ws + 1) * (NP * ROWS) + rem] : 0u;
            uint32_t s = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
            v ^= s & m1;
        }
        if (m2) {
            uint64_t D2 = 2 * D;
            uint64_t ws = wd + (D2 >> 5);
            uint32_t sh = D2 & 31;
            uint32_t lo = ws < nwords ? in[ws * (NP * ROWS) + rem] : 0u;
            uint32_t hi = (sh && ws + 1 < nwords) ? in[(ws + 1) * (NP * ROWS) + rem] : 0u;
            uint32_t s = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
            v ^= s & m2;
        }
        out[idx] = v;
    }
}

// four rows at a time: the conversion acts identically on every row, so the LDS kernels move 16-byte quads
struct u4 {
    uint32_t x, y, z, w;
};
__device__ __forceinline__ u4 operator^(u4 a, u4 b) { return {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
__device__ __forceinline__ u4 operator|(u4 a, u4 b) { return {a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w}; }
__device__ __forceinline__ u4 operator&(u4 a, uint32_t m) { return {a.x & m, a.y & m, a.z & m, a.w & m}; }
__device__ __forceinline__ u4 operator>>(u4 a, uint32_t s) { return {a.x >> s, a.y >> s, a.z >> s, a.w >> s}; }
__device__ __forceinline__ u4 operator<<(u4 a, uint32_t s) { return {a.x << s, a.y << s, a.z << s, a.w << s}; }
__device__ __forceinline__ u4 &operator^=(u4 &a, u4 b) {
    a = a ^ b;
    return a;
}
__device__ __forceinline__ u4 &operator|=(u4 &a, u4 b) {
    a = a | b;
    return a;
}

// ------------------------------------------------------------------------------------------------
// K2b: all conversion phases whose chunk fits a tile of 2^tb (<= 2^16) elements, fused in LDS, in place.
// Tile = 2^(tb-5) words x 16 rows of one plane (128 KiB at tb = 16). Each phase is split in the two
// hazard-free steps of the Taylor expansion at y = x^q + x on a chunk [lo | H]:
//   fold : H[j] ^= H[j + D]      (j < pw, D = half - pw)        -- top pw entries of H onto its bottom
//   shift: lo[pw + j] ^= H[j]    (j < D)
// forward = fold then shift, inverse = shift then fold.
struct conv_phases {
    uint8_t n;
    uint8_t lchunk[48], lpw[48];  // log2(chunk), log2(pw); half = chunk / 2
};
#define CT_ROWS 16
#define CT_RQ 4  // row quads per tile
#define CT_THREADS 1024

__device__ __forceinline__ void ct_fold(u4 *lds, uint32_t tile_words, uint32_t lchunk, uint32_t lpw) {
    const uint32_t chunk_w = 1u << (lchunk - 5), half_w = chunk_w >> 1;
    const uint32_t nchunks = tile_words >> (lchunk - 5);
    if (lpw >= 5) {
        const uint32_t pw_w = 1u << (lpw - 5);
        const uint32_t items = nchunks * pw_w * CT_RQ;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it & (CT_RQ - 1), wi = it / CT_RQ;
            uint32_t c = wi >> (lpw - 5), d = wi & (pw_w - 1);
            uint32_t base = c * chunk_w + half_w;
            lds[(base + d) * CT_RQ + r] ^= lds[(base + half_w - pw_w + d) * CT_RQ + r];
        }
    } else {
        const uint32_t pw = 1u << lpw;
        const uint32_t items = nchunks * CT_RQ;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it & (CT_RQ - 1), c = it / CT_RQ;
            uint32_t base = c * chunk_w + half_w;
            lds[base * CT_RQ + r] ^= lds[(base + half_w - 1) * CT_RQ + r] >> (32 - pw);
        }
    }
}

__device__ __forceinline__ void ct_shift(u4 *lds, uint32_t tile_words, uint32_t lchunk, uint32_t lpw) {
    const uint32_t chunk_w = 1u << (lchunk - 5), half_w = chunk_w >> 1;
    const uint32_t items = (tile_words >> 1) * CT_RQ;
    if (lpw >= 5) {
        const uint32_t pw_w = 1u << (lpw - 5);
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it & (CT_RQ - 1), wi = it / CT_RQ;
            uint32_t c = wi >> (lchunk - 6), d = wi & (half_w - 1);
            if (d >= pw_w) {
                uint32_t base = c * chunk_w;
                lds[(base + d) * CT_RQ + r] ^= lds[(base + half_w + d - pw_w) * CT_RQ + r];
            }
        }
    } else {
        const uint32_t pw = 1u << lpw;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it & (CT_RQ - 1), wi = it / CT_RQ;
            uint32_t c = wi >> (lchunk - 6), d = wi & (half_w - 1);
            uint32_t base = c * chunk_w;
            u4 h = lds[(base + half_w + d) * CT_RQ + r] << pw;
            if (d) h |= lds[(base + half_w + d - 1) * CT_RQ + r] >> (32 - pw);
            lds[(base + d) * CT_RQ + r] ^= h;
        }
    }
}

// chunk <= 32: every chunk lives inside one word; combined formula on the word itself
__device__ __forceinline__ void ct_intraword(u4 *lds, uint32_t tile_words, uint32_t lchunk, uint32_t lpw,
                                             bool inverse) {
    const uint32_t chunk = 1u << lchunk, pw = 1u << lpw, half = chunk >> 1, D = half - pw;
    uint32_t m1 = range_mask(half, half + pw) | range_mask(pw, half);
    uint32_t m2 = inverse ? 0u : range_mask(pw, 2 * pw);
    for (uint32_t s = chunk; s < 32; s <<= 1) {
        m1 |= m1 << s;
        m2 |= m2 << s;
    }
    const uint32_t items = tile_words * CT_RQ;
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        u4 v = lds[it];
        u4 o = v ^ ((v >> D) & m1);
        if (2 * D < 32) o ^= ((v >> (2 * D)) & m2);
        lds[it] = o;
    }
}

template <int NP>
__global__ void __launch_bounds__(CT_THREADS)
    k_conv_tile(uint32_t *__restrict__ D, uint32_t tb, conv_phases ph, int inverse) {
    __shared__ u4 lds[2048 * CT_RQ];
    const uint32_t tile_words = 1u << (tb - 5);
    // blockIdx.x = ((tile * NP) + plane) * 8 + rowgrp
    const uint32_t rowgrp = blockIdx.x & 7;
    const uint32_t plane = (blockIdx.x >> 3) % NP;
    const uint64_t tile = (blockIdx.x >> 3) / NP;
    uint32_t *g = D + (tile * tile_words * NP + plane) * ROWS + rowgrp * CT_ROWS;
    const uint32_t items = tile_words * CT_RQ;
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        uint32_t r = it & (CT_RQ - 1), wl = it / CT_RQ;
        lds[it] = *(const u4 *)(g + (uint64_t)wl * NP * ROWS + 4 * r);
    }
    __syncthreads();
    for (int q = 0; q < ph.n; q++) {
        const int i = inverse ? ph.n - 1 - q : q;
        const uint32_t lchunk = ph.lchunk[i], lpw = ph.lpw[i];
        if (lchunk <= 5) {
            ct_intraword(lds, tile_words, lchunk, lpw, inverse);
        } else if (!inverse) {
            ct_fold(lds, tile_words, lchunk, lpw);
            __syncthreads();
            ct_shift(lds, tile_words, lchunk, lpw);
        } else {
            ct_shift(lds, tile_words, lchunk, lpw);
            __syncthreads();
            ct_fold(lds, tile_words, lchunk, lpw);
        }
        __syncthreads();
    }
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        uint32_t r = it & (CT_RQ - 1), wl = it / CT_RQ;
        *(u4 *)(g + (uint64_t)wl * NP * ROWS + 4 * r) = lds[it];
    }
}

// ------------------------------------------------------------------------------------------------
// K2c: all conversion phases whose chunk spans several 2^16-element units, fused in LDS, in place.
// Tile = every unit u (NU = 2^(ell-16)) x NC columns x 32 rows of one plane; column 0 is a halo: the word that
// linearly precedes column 1 (offset o0 - 1, or offset 2047 of the same unit when o0 == 0, "wrap"). Two kinds of
// phases occur here (gf2fft_model.conversion_phases): pw >= 2^16 (whole units move: plain word XORs between
// units) and pw < 32 (a sub-word shift along the linear element index). The five sub-word phases shift by at
// most 31 bits in total, so one halo word per unit supplies every bit that crosses into the tile; its low bits
// decay phase by phase and it is never written back.
#define TOPK_ROWS 32
#define TOPK_RQ 8  // row quads
__device__ __forceinline__ void top_fold(u4 *lds, uint32_t NU, uint32_t NC, bool wrap, uint32_t lchunk,
                                         uint32_t lpw) {
    const uint32_t lcr = 31 - __clz(NC * TOPK_RQ);  // NC and TOPK_ROWS are powers of two
    const uint32_t chunk_u = 1u << (lchunk - 16), half_u = chunk_u >> 1, nchunks = NU >> (lchunk - 16);
    if (lpw >= 16) {
        const uint32_t pw_u = 1u << (lpw - 16);
        const uint32_t items = nchunks * pw_u * NC * TOPK_RQ;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t cr = it & ((1u << lcr) - 1), uj = it >> lcr;
            uint32_t c = uj >> (lpw - 16), j = uj & (pw_u - 1);
            uint32_t ud = c * chunk_u + half_u + j, us = ud + half_u - pw_u;
            lds[ud * NC * TOPK_RQ + cr] ^= lds[us * NC * TOPK_RQ + cr];
        }
    } else if (wrap) {
        const uint32_t pw = 1u << lpw;
        const uint32_t items = nchunks * TOPK_RQ;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it % TOPK_RQ, c = it / TOPK_RQ;
            uint32_t ud = c * chunk_u + half_u, us = c * chunk_u + chunk_u - 1;
            lds[(ud * NC + 1) * TOPK_RQ + r] ^= lds[(us * NC + 0) * TOPK_RQ + r] >> (32 - pw);
        }
    }
}

__device__ __forceinline__ void top_shift(u4 *lds, uint32_t NU, uint32_t NC, bool wrap, uint32_t lchunk,
                                          uint32_t lpw) {
    const uint32_t lcr = 31 - __clz(NC * TOPK_RQ);
    const uint32_t lhalf = lchunk - 17;
    const uint32_t chunk_u = 1u << (lchunk - 16), half_u = chunk_u >> 1;
    const uint32_t items = (NU / 2) * NC * TOPK_RQ;
    if (lpw >= 16) {
        const uint32_t pw_u = 1u << (lpw - 16);
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t cr = it & ((1u << lcr) - 1), ui = it >> lcr;
            uint32_t c = ui >> lhalf, d = ui & (half_u - 1);
            if (d >= pw_u) {
                uint32_t ud = c * chunk_u + d, us = c * chunk_u + half_u + d - pw_u;
                lds[ud * NC * TOPK_RQ + cr] ^= lds[us * NC * TOPK_RQ + cr];
            }
        }
    } else {
        const uint32_t pw = 1u << lpw;
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            uint32_t r = it % TOPK_RQ, col = (it / TOPK_RQ) & (NC - 1), ui = it >> lcr;
            uint32_t c = ui >> lhalf, d = ui & (half_u - 1);
            uint32_t ud = c * chunk_u + d, us = ud + half_u;
            u4 h = lds[(us * NC + col) * TOPK_RQ + r] << pw;
            if (col >= 2) {
                h |= lds[(us * NC + col - 1) * TOPK_RQ + r] >> (32 - pw);
            } else if (col == 1) {
                if (!wrap)
                    h |= lds[(us * NC + 0) * TOPK_RQ + r] >> (32 - pw);
                else if (d != 0)  // offset 0: the preceding word is offset 2047 of the previous unit
                    h |= lds[((us - 1) * NC + 0) * TOPK_RQ + r] >> (32 - pw);
            }
            lds[(ud * NC + col) * TOPK_RQ + r] ^= h;
        }
    }
}

template <int NP>
__global__ void __launch_bounds__(CT_THREADS)
    k_conv_top(const uint32_t *__restrict__ Din, uint32_t *__restrict__ Dout, uint32_t lnu, uint32_t lnc,
               conv_phases ph, int inverse) {
    __shared__ u4 lds[8192 + 256];
    const uint32_t NU = 1u << lnu, NC = 1u << lnc;
    const uint32_t W = NC - 1;  // payload columns
    // blockIdx.x = ((colblock * NP) + plane) * 4 + rowgrp
    const uint32_t rowgrp = blockIdx.x & 3;
    const uint32_t plane = (blockIdx.x >> 2) % NP;
    const uint32_t cb = (blockIdx.x >> 2) / NP;
    const uint32_t o0 = cb * W;
    const bool wrap = o0 == 0;
    // out of place: neighbouring tiles read each other's edge words as halos
    const uint32_t *g = Din + (uint64_t)plane * ROWS + rowgrp * TOPK_ROWS;
    uint32_t *go = Dout + (uint64_t)plane * ROWS + rowgrp * TOPK_ROWS;
    const uint32_t items = NU * NC * TOPK_RQ;
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        uint32_t r = it % TOPK_RQ, col = (it / TOPK_RQ) & (NC - 1), u = it >> (lnc + 3);
        uint32_t o = col ? o0 + col - 1 : (wrap ? 2047u : o0 - 1);
        u4 v = {0, 0, 0, 0};
        if (o < 2048) v = *(const u4 *)(g + ((uint64_t)u * 2048 + o) * NP * ROWS + 4 * r);
        lds[it] = v;
    }
    __syncthreads();
    for (int q = 0; q < ph.n; q++) {
        const int i = inverse ? ph.n - 1 - q : q;
        const uint32_t lchunk = ph.lchunk[i], lpw = ph.lpw[i];
        if (!inverse) {
            top_fold(lds, NU, NC, wrap, lchunk, lpw);
            __syncthreads();
            top_shift(lds, NU, NC, wrap, lchunk, lpw);
        } else {
            top_shift(lds, NU, NC, wrap, lchunk, lpw);
            __syncthreads();
            top_fold(lds, NU, NC, wrap, lchunk, lpw);
        }
        __syncthreads();
    }
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        uint32_t r = it % TOPK_RQ, col = (it / TOPK_RQ) & (NC - 1), u = it >> (lnc + 3);
        if (col == 0) continue;
        uint32_t o = o0 + col - 1;
        if (o < 2048) *(u4 *)(go + ((uint64_t)u * 2048 + o) * NP * ROWS + 4 * r) = lds[it];
    }
}

// ------------------------------------------------------------------------------------------------
// twiddle matrices. TW[tbl][idx][k] = row k of "multiply by point(2 * (idx << 10 tbl))";
// TWC[tbl][idx] = that point. Rows/points of a block index j are the XOR over tbl of entry (j >> 10 tbl) & 1023.
#define TW_TABLES 3
__global__ void k_build_tw(uint32_t *__restrict__ TW, uint32_t *__restrict__ TWC) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // (tbl, e, k)
    if (idx >= TW_TABLES * 1024 * 32) return;
    uint32_t k = idx & 31, e = (idx >> 5) & 1023, tbl = idx >> 15;
    uint64_t p = 2ull * ((uint64_t)e << (10 * tbl));
    uint32_t c = 0;
    for (int i = 0; i < 32; i++)
        if ((p >> i) & 1) c ^= d_BETA[i];
    uint32_t row = 0;
    for (int t = 0; t < 32; t++)
        if ((c >> t) & 1) row ^= d_RTAB[t * 32 + k];
    TW[idx] = row;
    if (k == 0) TWC[tbl * 1024 + e] = c;
}

struct twid {
    uint32_t c;        // the field element (nimber)
    uint32_t row[32];  // row k: bit j set <=> output bit k takes input bit j
};

__device__ __forceinline__ void load_twid(const uint32_t *__restrict__ TW, const uint32_t *__restrict__ TWC, uint32_t j,
                                          twid &tw) {
    // j is wave-uniform: everything below lives in SGPRs
    const uint32_t e0 = j & 1023, e1 = (j >> 10) & 1023, e2 = (j >> 20) & 1023;
    tw.c = TWC[e0] ^ TWC[1024 + e1] ^ TWC[2048 + e2];
#pragma unroll
    for (int k = 0; k < 32; k++)
        tw.row[k] = TW[e0 * 32 + k] ^ TW[(1024 + e1) * 32 + k] ^ TW[(2048 + e2) * 32 + k];
}

// t = c (x) y on two independent bit-sliced operands (rows l and l+64); the matrix is block diagonal
// with identical BxB blocks when c lies in the subfield GF(2^B)
template <int B>
__device__ __forceinline__ void mul_blocks(const twid &tw, const uint32_t y0[32], const uint32_t y1[32], uint32_t t0[32],
                                           uint32_t t1[32]) {
    // KB outputs at a time, input index outermost: 2*KB independent accumulation chains in flight
    constexpr int KB = 4;
#pragma unroll
    for (int blk = 0; blk < 32 / B; blk++) {
#pragma unroll
        for (int k0 = 0; k0 < B; k0 += KB) {
            uint32_t a0[KB], a1[KB];
#pragma unroll
            for (int kk = 0; kk < KB; kk++) a0[kk] = a1[kk] = 0;
#pragma unroll
            for (int j = 0; j < B; j++) {
#pragma unroll
                for (int kk = 0; kk < KB; kk++) {
                    const uint32_t m = 0u - ((tw.row[k0 + kk] >> j) & 1u);  // wave-uniform: s_bfe_i32
                    a0[kk] = XAND(a0[kk], y0[blk * B + j], m);
                    a1[kk] = XAND(a1[kk], y1[blk * B + j], m);
                }
            }
#pragma unroll
            for (int kk = 0; kk < KB; kk++) {
                t0[blk * B + k0 + kk] = a0[kk];
                t1[blk * B + k0 + kk] = a1[kk];
            }
        }
    }
}

__device__ __forceinline__ void mul_uniform(const twid &tw, const uint32_t y0[32], const uint32_t y1[32], uint32_t t0[32],
                                            uint32_t t1[32]) {
    const uint32_t c = tw.c;
    if (c < (1u << 8)) {
        mul_blocks<8>(tw, y0, y1, t0, t1);
    } else if (c < (1u << 16)) {
        mul_blocks<16>(tw, y0, y1, t0, t1);
    } else {
        mul_blocks<32>(tw, y0, y1, t0, t1);
    }
}

// ------------------------------------------------------------------------------------------------
// K3: Kronecker pack (16 polynomial-coefficient planes -> 32 nimber-coordinate planes, element = sum b_t g^t)
// fused with butterfly level ell-1, whose upper input half is zero (twiddle point(0) = 0): both halves = x.
__global__ void __launch_bounds__(256)
    k_qc_expand(const uint32_t *__restrict__ X, uint32_t *__restrict__ D, uint64_t nw_in) {
    static constexpr uint32_t GROW[32] = {
#include "qc_grow.inc"
    };
    const uint64_t total = nw_in * ROWS;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = idx / ROWS;
        const uint32_t row = idx % ROWS;
        uint32_t x[16];
#pragma unroll
        for (int t = 0; t < 16; t++) x[t] = X[(w * 16 + t) * ROWS + row];
#pragma unroll
        for (int k = 0; k < 32; k++) {
            uint32_t y = 0;
#pragma unroll
            for (int t = 0; t < 16; t++)
                if ((GROW[k] >> t) & 1) y ^= x[t];
            D[(w * 32 + k) * ROWS + row] = y;
            D[((w + nw_in) * 32 + k) * ROWS + row] = y;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4: one butterfly level i >= 5 (pairs of words at distance 2^(i-5)), forward:  x ^= tw*y; y ^= x
//                                                                        inverse:  y ^= x; x ^= tw*y
// One wave per word pair; lane l handles rows l and l+64.
template <bool INV>
__global__ void __launch_bounds__(256)
    k_fft_level(uint32_t *__restrict__ D, uint64_t npairs, uint32_t d /* = i - 5 */, const uint32_t *__restrict__ TW,
                const uint32_t *__restrict__ TWC) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t pi = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pi >= npairs) return;
    pi = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(pi >> 32)) << 32) |
         __builtin_amdgcn_readfirstlane((uint32_t)pi);
    const uint64_t w0 = ((pi >> d) << (d + 1)) | (pi & (((uint64_t)1 << d) - 1));
    const uint64_t w1 = w0 + ((uint64_t)1 << d);
    const uint32_t j = (uint32_t)(pi >> d);
    twid tw;
    load_twid(TW, TWC, j, tw);
    uint32_t *px = D + w0 * 32 * ROWS + lane;
    uint32_t *py = D + w1 * 32 * ROWS + lane;
    uint32_t x0[32], x1[32], y0[32], y1[32], t0[32], t1[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        x0[k] = px[k * ROWS];
        x1[k] = px[k * ROWS + 64];
        y0[k] = py[k * ROWS];
        y1[k] = py[k * ROWS + 64];
    }
    if (INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            y0[k] ^= x0[k];
            y1[k] ^= x1[k];
        }
    }
    if (tw.c != 0) {
        mul_uniform(tw, y0, y1, t0, t1);
#pragma unroll
        for (int k = 0; k < 32; k++) {
            x0[k] ^= t0[k];
            x1[k] ^= t1[k];
        }
    }
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            y0[k] ^= x0[k];
            y1[k] ^= x1[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        px[k * ROWS] = x0[k];
        px[k * ROWS + 64] = x1[k];
        py[k * ROWS] = y0[k];
        py[k * ROWS + 64] = y1[k];
    }
}

// ------------------------------------------------------------------------------------------------
// K4b: up to four consecutive word-pair levels fused through LDS. Tile = 2^nlev words at word stride 2^d0
// (levels i = 5 + d0 .. 5 + d0 + nlev - 1) x 32 planes x 64 rows = 128 KiB at nlev = 4. One wave per
// butterfly pair and level (the twiddle is wave uniform), lane = row.
template <int B>
__device__ __forceinline__ void mul_blocks1(const twid &tw, const uint32_t y[32], uint32_t t[32]) {
    constexpr int KB = 8;
#pragma unroll
    for (int blk = 0; blk < 32 / B; blk++) {
#pragma unroll
        for (int k0 = 0; k0 < B; k0 += KB) {
            uint32_t a[KB];
#pragma unroll
            for (int kk = 0; kk < KB; kk++) a[kk] = 0;
#pragma unroll
            for (int j = 0; j < B; j++) {
#pragma unroll
                for (int kk = 0; kk < KB; kk++)
                    a[kk] = XAND(a[kk], y[blk * B + j], 0u - ((tw.row[k0 + kk] >> j) & 1u));
            }
#pragma unroll
            for (int kk = 0; kk < KB; kk++) t[blk * B + k0 + kk] = a[kk];
        }
    }
}

#define FT_ROWS 64
#define FT_THREADS 512
template <bool INV>
__global__ void __launch_bounds__(FT_THREADS)
    k_fft_tile(uint32_t *__restrict__ D, uint32_t d0, uint32_t nlev, const uint32_t *__restrict__ TW,
               const uint32_t *__restrict__ TWC) {
    __shared__ uint32_t lds[16 * 32 * FT_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwt = 1u << nlev;
    const uint32_t rh = blockIdx.x & 1;
    const uint64_t ti = blockIdx.x >> 1;
    const uint64_t wbase = ((ti >> d0) << (d0 + nlev)) | (ti & (((uint64_t)1 << d0) - 1));
    uint32_t *g = D + rh * FT_ROWS + lane;
    // load: (wl, k) segments of 64 rows
    for (uint32_t s = wave; s < nwt * 32; s += FT_THREADS / 64) {
        uint32_t wl = s >> 5, k = s & 31;
        lds[s * FT_ROWS + lane] = g[((wbase + ((uint64_t)wl << d0)) * 32 + k) * ROWS];
    }
    __syncthreads();
    for (uint32_t q = 0; q < nlev; q++) {
        const uint32_t lv = INV ? q : nlev - 1 - q;
        for (uint32_t pr = wave; pr < nwt / 2; pr += FT_THREADS / 64) {
            const uint32_t a = ((pr >> lv) << (lv + 1)) | (pr & ((1u << lv) - 1));
            const uint32_t b = a + (1u << lv);
            const uint64_t w0 = wbase + ((uint64_t)a << d0);
            const uint32_t j = (uint32_t)(w0 >> (d0 + lv + 1));
            twid tw;
            load_twid(TW, TWC, j, tw);
            uint32_t *px = lds + (a * 32) * FT_ROWS + lane;
            uint32_t *py = lds + (b * 32) * FT_ROWS + lane;
            uint32_t x[32], y[32], t[32];
#pragma unroll
            for (int k = 0; k < 32; k++) {
                x[k] = px[k * FT_ROWS];
                y[k] = py[k * FT_ROWS];
            }
            if (INV) {
#pragma unroll
                for (int k = 0; k < 32; k++) y[k] ^= x[k];
            }
            if (tw.c != 0) {
                if (tw.c < (1u << 8))
                    mul_blocks1<8>(tw, y, t);
                else if (tw.c < (1u << 16))
                    mul_blocks1<16>(tw, y, t);
                else
                    mul_blocks1<32>(tw, y, t);
#pragma unroll
                for (int k = 0; k < 32; k++) x[k] ^= t[k];
            }
            if (!INV) {
#pragma unroll
                for (int k = 0; k < 32; k++) y[k] ^= x[k];
            }
#pragma unroll
            for (int k = 0; k < 32; k++) {
                px[k * FT_ROWS] = x[k];
                py[k * FT_ROWS] = y[k];
            }
        }
        __syncthreads();
    }
    for (uint32_t s = wave; s < nwt * 32; s += FT_THREADS / 64) {
        uint32_t wl = s >> 5, k = s & 31;
        g[((wbase + ((uint64_t)wl << d0)) * 32 + k) * ROWS] = lds[s * FT_ROWS + lane];
    }
}

// K4c: same tile, 16 waves: every butterfly is split over two waves by plane half. Valid while every twiddle of
// the pass lies in GF(2^16) (then multiply-by-twiddle acts on planes 0..15 and 16..31 independently with the
// same 16x16 matrix): levels i >= ell - 16. Twice the waves per SIMD of k_fft_tile at the same LDS footprint.
#define FT2_THREADS 1024
template <bool INV>
__global__ void __launch_bounds__(FT2_THREADS)
    k_fft_tile2(uint32_t *__restrict__ D, uint32_t d0, uint32_t nlev, const uint32_t *__restrict__ TW,
                const uint32_t *__restrict__ TWC) {
    __shared__ uint32_t lds[16 * 32 * FT_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwt = 1u << nlev;
    const uint32_t rh = blockIdx.x & 1;
    const uint64_t ti = blockIdx.x >> 1;
    const uint64_t wbase = ((ti >> d0) << (d0 + nlev)) | (ti & (((uint64_t)1 << d0) - 1));
    uint32_t *g = D + rh * FT_ROWS + lane;
    for (uint32_t s = wave; s < nwt * 32; s += FT2_THREADS / 64) {
        uint32_t wl = s >> 5, k = s & 31;
        lds[s * FT_ROWS + lane] = g[((wbase + ((uint64_t)wl << d0)) * 32 + k) * ROWS];
    }
    __syncthreads();
    for (uint32_t q = 0; q < nlev; q++) {
        const uint32_t lv = INV ? q : nlev - 1 - q;
        for (uint32_t itx = wave; itx < nwt; itx += FT2_THREADS / 64) {
            const uint32_t pr = itx >> 1, hf = itx & 1;
            const uint32_t a = ((pr >> lv) << (lv + 1)) | (pr & ((1u << lv) - 1));
            const uint32_t b = a + (1u << lv);
            const uint64_t w0 = wbase + ((uint64_t)a << d0);
            const uint32_t j = (uint32_t)(w0 >> (d0 + lv + 1));
            // 16x16 twiddle matrix rows (c < 2^16: rows 0..15, low 16 bits)
            const uint32_t e0 = j & 1023, e1 = (j >> 10) & 1023, e2 = (j >> 20) & 1023;
            const uint32_t c = TWC[e0] ^ TWC[1024 + e1] ^ TWC[2048 + e2];
            uint32_t row[16];
#pragma unroll
            for (int k = 0; k < 16; k++)
                row[k] = TW[e0 * 32 + k] ^ TW[(1024 + e1) * 32 + k] ^ TW[(2048 + e2) * 32 + k];
            uint32_t *px = lds + (a * 32 + hf * 16) * FT_ROWS + lane;
            uint32_t *py = lds + (b * 32 + hf * 16) * FT_ROWS + lane;
            uint32_t x[16], y[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                x[k] = px[k * FT_ROWS];
                y[k] = py[k * FT_ROWS];
            }
            if (INV) {
#pragma unroll
                for (int k = 0; k < 16; k++) y[k] ^= x[k];
            }
            if (c != 0) {
                if (c < (1u << 8)) {
#pragma unroll
                    for (int blk = 0; blk < 2; blk++) {
                        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int jj = 0; jj < 8; jj++)
#pragma unroll
                            for (int kk = 0; kk < 8; kk++)
                                acc[kk] = XAND(acc[kk], y[blk * 8 + jj], 0u - ((row[kk] >> jj) & 1u));
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) x[blk * 8 + kk] ^= acc[kk];
                    }
                } else {
#pragma unroll
                    for (int k0 = 0; k0 < 16; k0 += 8) {
                        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int jj = 0; jj < 16; jj++)
#pragma unroll
                            for (int kk = 0; kk < 8; kk++)
                                acc[kk] = XAND(acc[kk], y[jj], 0u - ((row[k0 + kk] >> jj) & 1u));
#pragma unroll
                        for (int kk = 0; kk < 8; kk++) x[k0 + kk] ^= acc[kk];
                    }
                }
            }
            if (!INV) {
#pragma unroll
                for (int k = 0; k < 16; k++) y[k] ^= x[k];
            }
#pragma unroll
            for (int k = 0; k < 16; k++) {
                px[k * FT_ROWS] = x[k];
                py[k * FT_ROWS] = y[k];
            }
        }
        __syncthreads();
    }
    for (uint32_t s = wave; s < nwt * 32; s += FT2_THREADS / 64) {
        uint32_t wl = s >> 5, k = s & 31;
        g[((wbase + ((uint64_t)wl << d0)) * 32 + k) * ROWS] = lds[s * FT_ROWS + lane];
    }
}

// ------------------------------------------------------------------------------------------------
// K5: the five intra-word levels (4..0 forward, 0..4 inverse), optionally with the pointwise multiply by
// FFT(a) in between. Level i pairs bit positions p and p + 2^i; the twiddle of position p is
// point(2*((32w+p) >> (i+1))) = point(w << (5-i)) ^ point(2*(p >> (i+1))): a wave-uniform part and a
// per-position part that lies in GF(2^8) (compile-time masks QC_MPOS).
__device__ __forceinline__ uint32_t lowmask(int i) {
    return i == 0 ? 0x55555555u : i == 1 ? 0x33333333u : i == 2 ? 0x0F0F0F0Fu : i == 3 ? 0x00FF00FFu : 0x0000FFFFu;
}

template <int I, bool INV>
__device__ __forceinline__ void low_level(uint32_t W0[32], uint32_t W1[32], uint64_t w, const uint32_t *__restrict__ TW,
                                          const uint32_t *__restrict__ TWC) {
    const uint32_t A = lowmask(I);
    constexpr int SH = 1 << I;
    uint32_t v0[32], v1[32], t0[32], t1[32];
    if (INV) {  // y ^= x first
#pragma unroll
        for (int k = 0; k < 32; k++) {
            W0[k] ^= (W0[k] & A) << SH;
            W1[k] ^= (W1[k] & A) << SH;
        }
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        v0[k] = (W0[k] >> SH) & A;
        v1[k] = (W1[k] >> SH) & A;
    }
    // uniform part: block index of the word's first position at this level
    const uint32_t ju = (uint32_t)(w << (4 - I));
    twid tw;
    load_twid(TW, TWC, ju, tw);
    if (tw.c != 0) {
        mul_uniform(tw, v0, v1, t0, t1);
    } else {
#pragma unroll
        for (int k = 0; k < 32; k++) t0[k] = t1[k] = 0;
    }
    if (I < 4) {  // per-position part, GF(2^8) blocks
#pragma unroll
        for (int blk = 0; blk < 4; blk++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t a0 = 0, a1 = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const uint32_t m = d_MPOS[(I * 8 + k) * 8 + j];
                    a0 = XAND(a0, v0[blk * 8 + j], m);
                    a1 = XAND(a1, v1[blk * 8 + j], m);
                }
                t0[blk * 8 + k] ^= a0;
                t1[blk * 8 + k] ^= a1;
            }
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        W0[k] ^= t0[k];  // t only has bits at the x positions
        W1[k] ^= t1[k];
    }
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) {
            W0[k] ^= (W0[k] & A) << SH;
            W1[k] ^= (W1[k] & A) << SH;
        }
    }
}

// MODE 0: forward low levels only (used for FFT(a)); MODE 1: forward, multiply by FFT(a), inverse
template <int MODE>
__global__ void __launch_bounds__(256)
    k_fft_low(uint32_t *__restrict__ D, uint64_t nw, const uint32_t *__restrict__ TW, const uint32_t *__restrict__ TWC,
              const uint32_t *__restrict__ Amat) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t w = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nw) return;
    w = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(w >> 32)) << 32) |
        __builtin_amdgcn_readfirstlane((uint32_t)w);
    uint32_t *p = D + w * 32 * ROWS + lane;
    uint32_t W0[32], W1[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        W0[k] = p[k * ROWS];
        W1[k] = p[k * ROWS + 64];
    }
    low_level<4, false>(W0, W1, w, TW, TWC);
    low_level<3, false>(W0, W1, w, TW, TWC);
    low_level<2, false>(W0, W1, w, TW, TWC);
    low_level<1, false>(W0, W1, w, TW, TWC);
    low_level<0, false>(W0, W1, w, TW, TWC);
    if (MODE == 1) {
        const uint32_t *M = Amat + w * 1024;  // wave-uniform address
        uint32_t t0[32], t1[32];
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 4) {
            uint32_t a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 32; j++) {
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const uint32_t m = M[(k0 + kk) * 32 + j];
                    a0[kk] = XAND(a0[kk], W0[j], m);
                    a1[kk] = XAND(a1[kk], W1[j], m);
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                t0[k0 + kk] = a0[kk];
                t1[k0 + kk] = a1[kk];
            }
        }
#pragma unroll
        for (int k = 0; k < 32; k++) {
            W0[k] = t0[k];
            W1[k] = t1[k];
        }
        low_level<0, true>(W0, W1, w, TW, TWC);
        low_level<1, true>(W0, W1, w, TW, TWC);
        low_level<2, true>(W0, W1, w, TW, TWC);
        low_level<3, true>(W0, W1, w, TW, TWC);
        low_level<4, true>(W0, W1, w, TW, TWC);
    }
#pragma unroll
    for (int k = 0; k < 32; k++) {
        p[k * ROWS] = W0[k];
        p[k * ROWS + 64] = W1[k];
    }
}

// ---- single-row variant: one row per lane (half the registers: 4+ waves per SIMD keep both the VALU and the
// scalar port that extracts the masks busy)
template <int B, int I>
__device__ __forceinline__ void low_mul_acc(const twid &tw, const uint32_t v[32], uint32_t W[32]) {
    constexpr int KB = 8;
#pragma unroll
    for (int blk = 0; blk < 32 / B; blk++) {
#pragma unroll
        for (int k0 = 0; k0 < B; k0 += KB) {
            uint32_t a[KB];
#pragma unroll
            for (int kk = 0; kk < KB; kk++) a[kk] = 0;
#pragma unroll
            for (int j = 0; j < B; j++) {
#pragma unroll
                for (int kk = 0; kk < KB; kk++)
                    a[kk] = XAND(a[kk], v[blk * B + j], 0u - ((tw.row[k0 + kk] >> j) & 1u));
            }
#pragma unroll
            for (int kk = 0; kk < KB; kk++) W[blk * B + k0 + kk] ^= a[kk];
        }
    }
}

template <int I, bool INV>
__device__ __forceinline__ void low_level1(uint32_t W[32], uint64_t w, const uint32_t *__restrict__ TW,
                                           const uint32_t *__restrict__ TWC) {
    const uint32_t A = lowmask(I);
    constexpr int SH = 1 << I;
    uint32_t v[32];
    if (INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] ^= (W[k] & A) << SH;
    }
#pragma unroll
    for (int k = 0; k < 32; k++) v[k] = (W[k] >> SH) & A;  // y positions; the updates below touch x positions only
    const uint32_t ju = (uint32_t)(w << (4 - I));
    twid tw;
    load_twid(TW, TWC, ju, tw);
    if (tw.c != 0) {
        if (tw.c < (1u << 8))
            low_mul_acc<8, I>(tw, v, W);
        else if (tw.c < (1u << 16))
            low_mul_acc<16, I>(tw, v, W);
        else
            low_mul_acc<32, I>(tw, v, W);
    }
    if (I < 4) {
#pragma unroll
        for (int blk = 0; blk < 4; blk++)
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t a = 0;
#pragma unroll
                for (int j = 0; j < 8; j++) a = XAND(a, v[blk * 8 + j], d_MPOS[(I * 8 + k) * 8 + j]);
                W[blk * 8 + k] ^= a;
            }
    }
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] ^= (W[k] & A) << SH;
    }
}

template <int MODE>
__global__ void __launch_bounds__(256, 4)
    k_fft_low1(uint32_t *__restrict__ D, uint64_t nw, const uint32_t *__restrict__ TW, const uint32_t *__restrict__ TWC,
               const uint32_t *__restrict__ Amat) {
    const uint32_t lane = threadIdx.x & 63;
    uint64_t item = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (w, row half)
    if (item >= 2 * nw) return;
    item = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(item >> 32)) << 32) |
           __builtin_amdgcn_readfirstlane((uint32_t)item);
    const uint64_t w = item >> 1;
    uint32_t *p = D + w * 32 * ROWS + (item & 1) * 64 + lane;
    uint32_t W[32];
#pragma unroll
    for (int k = 0; k < 32; k++) W[k] = p[k * ROWS];
    low_level1<4, false>(W, w, TW, TWC);
    low_level1<3, false>(W, w, TW, TWC);
    low_level1<2, false>(W, w, TW, TWC);
    low_level1<1, false>(W, w, TW, TWC);
    low_level1<0, false>(W, w, TW, TWC);
    if (MODE == 1) {
        const uint32_t *M = Amat + w * 1024;  // wave-uniform address
        uint32_t t[32];
#pragma unroll
        for (int k0 = 0; k0 < 32; k0 += 8) {
            uint32_t a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 32; j++) {
#pragma unroll
                for (int kk = 0; kk < 8; kk++) a[kk] = XAND(a[kk], W[j], M[(k0 + kk) * 32 + j]);
            }
#pragma unroll
            for (int kk = 0; kk < 8; kk++) t[k0 + kk] = a[kk];
        }
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] = t[k];
        low_level1<0, true>(W, w, TW, TWC);
        low_level1<1, true>(W, w, TW, TWC);
        low_level1<2, true>(W, w, TW, TWC);
        low_level1<3, true>(W, w, TW, TWC);
        low_level1<4, true>(W, w, TW, TWC);
    }
#pragma unroll
    for (int k = 0; k < 32; k++) p[k * ROWS] = W[k];
}

// Amat[w][k][j] bit p = bit k of (FFT(a)[32 w + p] (x) 2^j); FFT(a) is row 0 of Da
__global__ void k_build_amat(const uint32_t *__restrict__ Da, uint64_t nw, uint32_t *__restrict__ Amat) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (w, k, j)
    if (idx >= nw * 1024) return;
    const uint64_t w = idx >> 10;
    const uint32_t k = (idx >> 5) & 31, j = idx & 31;
    uint32_t m = 0;
    for (int t = 0; t < 32; t++)
        if ((d_RTAB[t * 32 + k] >> j) & 1) m ^= Da[(w * 32 + t) * ROWS];
    Amat[idx] = m;
}

// ------------------------------------------------------------------------------------------------
// K7: nimber coordinates -> coefficients on 1, g, ..., g^31 (in place), all words
__global__ void __launch_bounds__(256) k_qc_unpack(uint32_t *__restrict__ D, uint64_t nw) {
    static constexpr uint32_t GINVROW[32] = {
#include "qc_ginvrow.inc"
    };
    const uint64_t total = nw * ROWS;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t w = idx / ROWS;
        const uint32_t row = idx % ROWS;
        uint32_t *p = D + w * 32 * ROWS + row;
        uint32_t x[32];
#pragma unroll
        for (int k = 0; k < 32; k++) x[k] = p[k * ROWS];
#pragma unroll
        for (int t = 0; t < 32; t++) {
            uint32_t y = 0;
#pragma unroll
            for (int k = 0; k < 32; k++)
                if ((GINVROW[t] >> k) & 1) y ^= x[k];
            p[t * ROWS] = y;
        }
    }
}

// K9: overlap-add of the 31-bit product elements and transpose back to Blocks:
// T[16*(32w+p) + t] bit row = C_t[w] ^ (C_{16+t}[w] << 1 | C_{16+t}[w-1] >> 31), bit p.
// One block = 2 consecutive words w; the slabs are staged through LDS (33-word padded groups of 32 rows)
// so that both the global reads and the Block stores are coalesced.
#define ST_W 2
__global__ void __launch_bounds__(128) k_qc_store(const uint32_t *__restrict__ D, uint64_t nw, blk_t *__restrict__ T) {
    __shared__ uint32_t slab[(ST_W * 128 + 64) * 33];  // [hi planes of w0-1 | w0 | w0+1], group = plane*4 + row/32
    __shared__ uint32_t tile[ST_W][512 * 4];
    const uint64_t w0 = (uint64_t)blockIdx.x * ST_W;
    // stage: groups 0..63 = planes 16..31 of w0-1; groups 64.. = all planes of w0, w0+1
    for (uint32_t i = threadIdx.x; i < (ST_W * 4096 + 2048) / 4; i += blockDim.x) {
        uint32_t x = 4 * i;  // word index in the staged range
        uint4 v = make_uint4(0, 0, 0, 0);
        if (x < 2048) {
            if (w0 > 0) v = *(const uint4 *)(D + ((w0 - 1) * 32 + 16) * ROWS + x);
        } else {
            uint64_t w = w0 + (x - 2048) / 4096;
            if (w < nw) v = *(const uint4 *)(D + w0 * 32 * ROWS + (x - 2048));
        }
        uint32_t g = x >> 5, b = x & 31;
        uint32_t *d = slab + g * 33 + b;
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t w = w0 + wave;
    if (w < nw) {
        const uint32_t t = lane >> 2, q = lane & 3;
        uint32_t a[32];
        const uint32_t *lo = slab + (64 + wave * 128 + t * 4 + q) * 33;
        const uint32_t *hi = slab + (64 + wave * 128 + (16 + t) * 4 + q) * 33;
        // previous word's hi planes: for wave 0 the staged w0-1 copy, else the preceding slab
        const uint32_t *hp = wave == 0 ? slab + (t * 4 + q) * 33 : slab + (64 + (wave - 1) * 128 + (16 + t) * 4 + q) * 33;
#pragma unroll
        for (int b = 0; b < 32; b++) a[b] = lo[b] ^ (hi[b] << 1) ^ (hp[b] >> 31);
        transpose32(a);
#pragma unroll
        for (int p = 0; p < 32; p++) tile[wave][(16 * p + t) * 4 + q] = a[p];
    }
    __syncthreads();
    if (w >= nw) return;
    for (uint32_t i = lane; i < 512; i += 64) T[w * 512 + i] = *(blk_t *)&tile[wave][i * 4];
}

// ------------------------------------------------------------------------------------------------
// K10: fold mod x^P - 1, add V0, add the tail terms:
//   B[k] = V0[k] ^ V0[k+P] ^ T[k] ^ T[k+P] ^ sum_{i: a_hi[i]} V1'[(k - S - i) mod P]
//                                         ^ sum_{j < tail} a_lo[(k - S - j) mod P] V1'[S + j]
// V0 = V[0 .. nb128), V1' = (V1 mod x^P - 1). a2 = a' repeated twice as a bit vector.
__device__ __forceinline__ blk_t ld_v1p(const blk_t *V1, uint64_t v1_len, uint64_t P, uint64_t c) {
    blk_t v = V1[c];
    if (c + P < v1_len) {
        blk_t u = V1[c + P];
        v.x ^= u.x;
        v.y ^= u.y;
        v.z ^= u.z;
        v.w ^= u.w;
    }
    return v;
}

__global__ void __launch_bounds__(256)
    k_qc_final(const blk_t *__restrict__ V, uint64_t nb128, uint64_t P, uint64_t S, uint64_t tail, uint64_t n,
               const blk_t *__restrict__ T, uint64_t t_len, const uint32_t *__restrict__ a2,
               const uint32_t *__restrict__ a_hi /* tail bits, packed */, blk_t *__restrict__ out) {
    const blk_t *V1 = V + nb128;
    for (uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (uint64_t)gridDim.x * blockDim.x) {
        blk_t r = V[k];
        if (k + P < nb128) {
            blk_t u = V[k + P];
            r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
        }
        for (uint64_t m = k; m < t_len; m += P) {
            blk_t u = T[m];
            r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
        }
        if (tail) {
            // q = (k - S) mod P
            uint64_t q = k >= S ? k - S : k + P - S;
            for (uint64_t i = 0; i < tail; i++) {
                if ((a_hi[i >> 5] >> (i & 31)) & 1u) {
                    uint64_t c = q >= i ? q - i : q + P - i;
                    blk_t u = ld_v1p(V1, nb128, P, c);
                    r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
                }
            }
            for (uint64_t j = 0; j < tail; j++) {
                uint64_t c = q >= j ? q - j : q + P - j;  // index into a'
                if (c < S) {
                    uint32_t bit = (a2[c >> 5] >> (c & 31)) & 1u;
                    if (bit) {
                        blk_t u = ld_v1p(V1, nb128, P, S + j);
                        r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
                    }
                }
            }
        }
        out[k] = r;
    }
}

// ================================================================================================
// host orchestration
// ================================================================================================
struct phase_t {
    uint64_t chunk, pw, half;
};

static void conv_rec(std::vector<phase_t> &ph, uint32_t nbits, uint64_t unit) {
    if (nbits <= 1) return;
    uint32_t t = 0;
    while ((1u << (t + 1)) < nbits) t++;
    uint32_t qb = 1u << t, R = nbits - qb;
    for (int pb = (int)R - 1; pb >= 0; pb--) {
        uint64_t pw = ((uint64_t)1 << pb) * unit;
        uint64_t half = ((uint64_t)1 << (pb + qb)) * unit;
        ph.push_back({2 * half, pw, half});
    }
    conv_rec(ph, R, unit << qb);
    conv_rec(ph, qb, unit);
}

static inline unsigned gridn(uint64_t n, unsigned block, unsigned cap = 0x7fffffffu) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

static inline uint32_t ilog2(uint64_t x) {
    uint32_t l = 0;
    while ((1ull << l) < x) l++;
    return l;
}

// runs the conversion on `bufs[cur]`; phases with chunk > 2^16 elements go through the generic ping-pong
// kernel, the rest through the fused LDS tile kernel (in place). Returns the buffer index holding the result.
template <int NP>
static int run_conversion(sot_ctx *c, uint32_t *bufs[2], int cur, uint32_t ell_elems, uint64_t nwords, bool inverse) {
    std::vector<phase_t> ph;
    conv_rec(ph, ell_elems, 1);
    const uint32_t tb = ell_elems < 16 ? ell_elems : 16;
    std::vector<phase_t> big;
    conv_phases small;
    small.n = 0;
    for (auto &p : ph) {
        if (p.chunk > (1ull << tb)) {
            big.push_back(p);  // conv_rec emits every chunk > 2^16 phase before the tile-local ones
        } else {
            small.lchunk[small.n] = (uint8_t)ilog2(p.chunk);
            small.lpw[small.n] = (uint8_t)ilog2(p.pw);
            small.n++;
        }
    }
    const uint64_t total = nwords * NP * ROWS;
    const unsigned grid = gridn(total, 256, 1u << 20);
    const unsigned tgrid = (unsigned)((nwords >> (tb - 5)) * NP * 8);
    const int kid = NP == 16 ? KID_CONV16 : KID_CONV32;
    const int kid_t = NP == 16 ? KID_CONV16_TILE : KID_CONV32_TILE;
    // the multi-unit phases fuse into k_conv_top when each is a whole-unit move or a sub-word shift
    conv_phases top;
    top.n = 0;
    bool top_ok = !big.empty() && ell_elems > 16 && ell_elems - 16 <= 7 && !getenv("SOT_NO_TOP");
    uint32_t shift_sum = 0;
    for (auto &p : big) {
        uint32_t lpw = ilog2(p.pw);
        if (lpw >= 5 && lpw < 16) top_ok = false;
        if (lpw < 5) shift_sum += 1u << lpw;
        top.lchunk[top.n] = (uint8_t)ilog2(p.chunk);
        top.lpw[top.n] = (uint8_t)lpw;
        top.n++;
    }
    if (shift_sum > 31) top_ok = false;
    const uint32_t lnu = top_ok ? ell_elems - 16 : 0;
    const uint32_t NC = top_ok ? 1024u >> lnu : 0;  // NU * NC * 32 rows = 32768 words
    const unsigned topgrid = top_ok ? (unsigned)(((2048 + NC - 2) / (NC - 1)) * NP * 4) : 0;
    if (!inverse) {
        if (top_ok) {
            SOT_LAUNCH(c, kid, k_conv_top<NP>, dim3(topgrid), dim3(CT_THREADS), bufs[cur], bufs[cur ^ 1], lnu, ilog2(NC), top, 0);
            cur ^= 1;
        } else {
            for (size_t i = 0; i < big.size(); i++) {
                SOT_LAUNCH(c, kid, k_conv_phase<NP>, dim3(grid), dim3(256), bufs[cur], bufs[cur ^ 1], nwords,
                           big[i].chunk, big[i].pw, big[i].half, 0);
                cur ^= 1;
            }
        }
        if (small.n) SOT_LAUNCH(c, kid_t, k_conv_tile<NP>, dim3(tgrid), dim3(CT_THREADS), bufs[cur], tb, small, 0);
    } else {
        if (small.n) SOT_LAUNCH(c, kid_t, k_conv_tile<NP>, dim3(tgrid), dim3(CT_THREADS), bufs[cur], tb, small, 1);
        if (top_ok) {
            SOT_LAUNCH(c, kid, k_conv_top<NP>, dim3(topgrid), dim3(CT_THREADS), bufs[cur], bufs[cur ^ 1], lnu, ilog2(NC), top, 1);
            cur ^= 1;
        } else {
            for (size_t i = big.size(); i-- > 0;) {
                SOT_LAUNCH(c, kid, k_conv_phase<NP>, dim3(grid), dim3(256), bufs[cur], bufs[cur ^ 1], nwords,
                           big[i].chunk, big[i].pw, big[i].half, 1);
                cur ^= 1;
            }
        }
    }
    return cur;
}

// forward transform of the 16-plane input in X[xcur] into D0 (levels ell-1 .. 5); low levels left to the caller
static int forward_to_D(sot_ctx *c, int xcur) {
    const qc_geom &g = c->g;
    uint32_t *xb[2] = {c->X0, c->X1};
    xcur = run_conversion<16>(c, xb, xcur, g.ell - 1, g.nw_in, false);
    SOT_LAUNCH(c, KID_QC_EXPAND, k_qc_expand, dim3(gridn(g.nw_in * ROWS, 256, 1u << 20)), dim3(256), xb[xcur],
                       c->D0, g.nw_in);
    // word-pair levels i = ell-2 .. 5 (d = i - 5), four per pass from the top
    for (int dtop = (int)g.ell - 7; dtop >= 0;) {
        int nlev = dtop + 1 < 4 ? dtop + 1 : 4;
        int d0 = dtop - nlev + 1;
        if (5 + d0 + 16 >= (int)g.ell)
            SOT_LAUNCH(c, KID_FFT_LEVEL, k_fft_tile2<false>, dim3((unsigned)((g.nw >> nlev) * 2)), dim3(FT2_THREADS),
                       c->D0, (uint32_t)d0, (uint32_t)nlev, c->TW, c->TWC);
        else
            SOT_LAUNCH(c, KID_FFT_LEVEL, k_fft_tile<false>, dim3((unsigned)((g.nw >> nlev) * 2)), dim3(FT_THREADS),
                       c->D0, (uint32_t)d0, (uint32_t)nlev, c->TW, c->TWC);
        dtop = d0 - 1;
    }
    SOT_HIP(hipGetLastError());
    return SOT_OK;
}

static int upload_a_and_transform(sot_ctx *c) {
    const sot_conf &cf = c->conf;
    const qc_geom &g = c->g;
    const uint64_t nb128 = cf.n_blocks * 128;
    // a = first n64 u64 of AesRng(seed = Block::from(1)) (quasi_cyclic_encode.rs:127-138)
    blk_t *d_araw = nullptr;
    SOT_HIP(hipMalloc((void **)&d_araw, cf.n_blocks * 16));
    uint8_t key[16] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    int rc = util_aes_ctr_device(c, key, 0, d_araw, cf.n_blocks);
    std::vector<uint32_t> a(cf.n_blocks * 4 + 2);
    if (rc == SOT_OK && hipMemcpy(a.data(), d_araw, cf.n_blocks * 16, hipMemcpyDeviceToHost) != hipSuccess)
        rc = SOT_EDEVICE;
    hipFree(d_araw);
    SOT_TRY(rc);
    // a' = a mod (x^P - 1): fold bits [P, nb128) onto [0, nb128 - P)
    auto getbit = [&](uint64_t i) -> uint32_t { return (a[i >> 5] >> (i & 31)) & 1u; };
    const uint64_t P = cf.P;
    std::vector<uint32_t> ap((P + 31) / 32 + 2, 0);
    for (uint64_t w = 0; w < (P + 31) / 32; w++) ap[w] = a[w];
    if (P & 31) ap[P / 32] &= (1u << (P & 31)) - 1u;
    for (uint64_t i = P; i < nb128; i++)
        if (getbit(i)) ap[(i - P) >> 5] ^= 1u << ((i - P) & 31);
    // tail bits a'[S .. P)
    c->a_hi_host.assign((g.tail + 31) / 32 + 1, 0);
    for (uint64_t i = 0; i < g.tail; i++)
        if ((ap[(g.S + i) >> 5] >> ((g.S + i) & 31)) & 1u) c->a_hi_host[i >> 5] |= 1u << (i & 31);
    SOT_HIP(hipMemcpy(c->a2, ap.data(), ap.size() * 4, hipMemcpyHostToDevice));
    SOT_HIP(hipMemcpy(c->a_hi, c->a_hi_host.data(), c->a_hi_host.size() * 4, hipMemcpyHostToDevice));
    // FFT(a_lo): row 0 of the plane layout
    SOT_HIP(hipMemsetAsync(c->X0, 0, g.nw_in * 16 * ROWS * 4, c->stream));
    hipLaunchKernelGGL(k_a_load, dim3(gridn(g.nw_in * 16, 256)), dim3(256), 0, c->stream, c->a2, g.lo_bits, g.nw_in,
                       c->X0);
    SOT_TRY(forward_to_D(c, 0));
    hipLaunchKernelGGL(k_fft_low<0>, dim3(gridn(g.nw, 4)), dim3(256), 0, c->stream, c->D0, g.nw, c->TW, c->TWC,
                       (const uint32_t *)nullptr);
    hipLaunchKernelGGL(k_build_amat, dim3(gridn(g.nw * 1024, 256)), dim3(256), 0, c->stream, c->D0, g.nw, c->Amat);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));
    return SOT_OK;
}

int qc_init(sot_ctx *c) {
    const sot_conf &cf = c->conf;
    qc_geom &g = c->g;
    const uint64_t nb128 = cf.n_blocks * 128;
    if (nb128 < cf.P) {
        sot_set_error("unsupported geometry: n_blocks*128 < P");
        return SOT_EINVAL;
    }
    // FFT capacity S = 8 L bits per operand; up to QC_TAIL_MAX bits above it go through the tail terms
    const uint64_t TAIL_MAX = 256;
    uint32_t ell = 11;
    while ((8ull << ell) + TAIL_MAX < cf.P) ell++;
    g.ell = ell;
    g.L = 1ull << ell;
    g.S = 8 * g.L;
    g.tail = cf.P > g.S ? cf.P - g.S : 0;
    g.lo_bits = cf.P < g.S ? cf.P : g.S;
    g.nw_in = g.L / 64;
    g.nw = g.L / 32;
    if (ell > 24) {
        sot_set_error("n too large for the twiddle tables");
        return SOT_EINVAL;
    }
    SOT_TRY(dev_alloc(c, (void **)&c->X0, g.nw_in * 16 * ROWS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->X1, g.nw_in * 16 * ROWS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->D0, g.nw * 32 * ROWS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->D1, g.nw * 32 * ROWS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->Amat, g.nw * 1024 * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->T, 2 * g.S * 16));
    SOT_TRY(dev_alloc(c, (void **)&c->a2, ((cf.P + 31) / 32 + 2) * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->a_hi, ((g.tail + 31) / 32 + 1) * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->TW, TW_TABLES * 1024 * 32 * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->TWC, TW_TABLES * 1024 * 4));
    hipLaunchKernelGGL(k_build_tw, dim3(TW_TABLES * 1024 * 32 / 256), dim3(256), 0, c->stream, c->TW, c->TWC);
    SOT_HIP(hipGetLastError());
    return upload_a_and_transform(c);
}

void qc_free(sot_ctx *c) {
    void *ptrs[] = {c->X0, c->X1, c->D0, c->D1, c->Amat, c->T, c->a2, c->a_hi, c->TW, c->TWC};
    for (void *p : ptrs)
        if (p) hipFree(p);
}

int qc_encode(sot_ctx *c, int role, const uint64_t *points, uint64_t npoints) {
    const sot_conf &cf = c->conf;
    const qc_geom &g = c->g;
    SOT_HIP(hipSetDevice(c->device));
    blk_t *V = c->role[role].V;
    const uint64_t nb128 = cf.n_blocks * 128;
    const uint64_t nV = cf.cols * 128;
    if (role == SOT_ROLE_RECEIVER) {
        SOT_HIP(hipMemcpyAsync(c->d_points, points, npoints * 8, hipMemcpyHostToDevice, c->stream));
        SOT_LAUNCH(c, KID_CLEAR_LSB, k_clear_lsb, dim3(gridn(nV, 256, 4096)), dim3(256), V, nV);
        SOT_LAUNCH(c, KID_CLEAR_LSB, k_set_points, dim3((npoints + 127) / 128), dim3(128), V, c->d_points,
                           (uint32_t)npoints, nV);
    }
    SOT_LAUNCH(c, KID_QC_LOAD, k_qc_load, dim3(gridn(g.nw_in, 4)), dim3(256), V + nb128, nb128, cf.P, g.lo_bits,
                       g.nw_in, c->X0);
    SOT_TRY(forward_to_D(c, 0));
    SOT_LAUNCH(c, KID_FFT_LOW_MUL, k_fft_low1<1>, dim3(gridn(2 * g.nw, 4)), dim3(256), c->D0, g.nw, c->TW, c->TWC,
                       c->Amat);
    for (int d0 = 0; d0 <= (int)g.ell - 6;) {
        int rem = (int)g.ell - 5 - d0;
        int nlev = rem < 4 ? rem : 4;
        if (5 + d0 + 16 >= (int)g.ell)
            SOT_LAUNCH(c, KID_IFFT_LEVEL, k_fft_tile2<true>, dim3((unsigned)((g.nw >> nlev) * 2)), dim3(FT2_THREADS),
                       c->D0, (uint32_t)d0, (uint32_t)nlev, c->TW, c->TWC);
        else
            SOT_LAUNCH(c, KID_IFFT_LEVEL, k_fft_tile<true>, dim3((unsigned)((g.nw >> nlev) * 2)), dim3(FT_THREADS),
                       c->D0, (uint32_t)d0, (uint32_t)nlev, c->TW, c->TWC);
        d0 += nlev;
    }
    SOT_LAUNCH(c, KID_QC_UNPACK, k_qc_unpack, dim3(gridn(g.nw * ROWS, 256, 1u << 20)), dim3(256), c->D0, g.nw);
    uint32_t *db[2] = {c->D0, c->D1};
    int cur = run_conversion<32>(c, db, 0, g.ell, g.nw, true);
    SOT_LAUNCH(c, KID_QC_STORE, k_qc_store, dim3(gridn(g.nw, ST_W)), dim3(64 * ST_W), db[cur], g.nw, c->T);
    SOT_LAUNCH(c, KID_QC_FINAL, k_qc_final, dim3(gridn(cf.n, 256, 8192)), dim3(256), V, nb128, cf.P, g.S, g.tail,
                       cf.n, c->T, 2 * g.S, c->a2, c->a_hi, c->role[role].enc);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));
    c->role[role].encoded = true;
    return SOT_OK;
}
