// QuasiCyclic dual-LPN encode on gfx950: B[k] = V[k] ^ V[k+P] ^ ((a * V1) mod (x^P - 1))[k]
//
// Reference: zappot/src/silent_ot/quasi_cyclic_encode.rs:29-151,262-344 (128 per-row products
// a * b1_r via bitpolymul's additive FFT over GF(2^128), + b0_r, modp, 128x128 transposes).
//
// MI355X formulation: the 128 rows multiply by the same public a, and a Block of the untransposed
// PPRF output V holds one coefficient of all 128 row polynomials, so the rows are processed bit-sliced:
// Z[w][row] is a 32-bit word holding coefficients 32w .. 32w+31 of row `row` (512 B "granule" per w).
// The product is a Frobenius-pruned additive FFT over GF(2^32) (nimbers, Cantor-basis points), executed
// IN PLACE on Z from the first basis-conversion pass to the last:
//   f in GF(2)[x], deg f < 2^m, is determined by its values on the coset beta_m + W_m; squaring permutes the
//   coset with orbits of size 32 and f(w^2) = f(w)^2, so one value per orbit is enough, and
//   {index bits m-1, m-2, m-4, m-8, m-16 = 0} is a cross-section. The Lin-Chung-Han butterfly network
//   therefore drops the "bit = 1" branch at levels r = m - i in {1, 2, 4, 8, 16}. At such a level the kept branch
//   is x + tau y with x, y in GF(2^r) and tau = 2^r + tau_lo (nimber), i.e. (x + tau_lo y) | (y << r): the
//   values double in width in place, and word-index bit b(r) = m - r - 5 becomes bit log2(r) of the plane
//   (nimber coordinate) index. The data stay at 2^m bits per row throughout (no Kronecker padding).
// Every field operation is a GF(2)-linear map on planes: v_bitop3 (a ^ (b & mask)) with wave-uniform masks.
// The math, in this exact layout, is executable in fafft_model.py (and gf2fft_model.py / gen_tables.py).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "sot_internal.h"

#define ROWS 128  // bit rows of a Block
#ifndef QC_EXP
// profiling experiments only (tools/exp.sh; results are wrong when set): 1 = k_fft_low without word levels,
// 2 = without intra-word levels, 4 = k_qc_finish without the tail sums, 8 = without the transposes, 16 = k_mid
// without levels, 64 = k_mid without the level barriers, 128 = k_top without the unit phases, 256 = without the levels
#define QC_EXP 0
#endif
#ifndef LOW_RING
#define LOW_RING 8  // k_fft_low: mask blocks (<= 960 B) in a value word's 8 KiB ring, LOW_RING - 1 of them loaded ahead
#endif

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
// The leaves V are never modified: the noise indicator lives in a bit vector sb (one bit per Block of V, set at the
// `parts` interleaved points) and every load of V on the receiver's encode path substitutes it for bit 0.
__global__ void k_set_points(uint32_t *__restrict__ sb, const uint64_t *__restrict__ pts, uint32_t npts, uint64_t nV) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npts && pts[i] < nV) atomicOr(sb + (pts[i] >> 5), 1u << (pts[i] & 31));
}
// V[idx], with bit `sbpos` := sb[idx] when sb != nullptr (receiver with packed choice bits: sbpos = 0, the receiver's
// bit row 0; in the slot Blocks of a sharded context the receiver's row 0 is slot 128 / world of rank 0, shard.hip)
// lw = words per stored leaf: 4 (whole Blocks), or 2 / 1 for the slot Blocks of a sharded context of 4 / 8 ranks, which
// hold 64 / 32 live bits and are stored packed (kernel-uniform)
__device__ __forceinline__ blk_t ld_leaf(const blk_t *__restrict__ V, uint64_t idx, const uint32_t *__restrict__ sb,
                                         uint32_t sbpos, uint32_t lw = 4) {
    blk_t v;
    if (lw == 4) {
        v = V[idx];
    } else if (lw == 2) {
        const uint2 t = ((const uint2 *)V)[idx];
        v = make_uint4(t.x, t.y, 0, 0);
    } else {
        v = make_uint4(((const uint32_t *)V)[idx], 0, 0, 0);
    }
    if (sb) {
        const uint32_t sh = sbpos & 31u, m = ~(1u << sh), b = ((sb[idx >> 5] >> (idx & 31)) & 1u) << sh;
        switch (sbpos >> 5) {  // kernel-uniform
            case 0: v.x = (v.x & m) | b; break;
            case 1: v.y = (v.y & m) | b; break;
            case 2: v.z = (v.z & m) | b; break;
            default: v.w = (v.w & m) | b; break;
        }
    }
    return v;
}

// receiver with UNPACKED choice bits (ChoiceBitPacking::False, quasi_cyclic_encode.rs:65-68,104-117): all 128 rows of A
// come from rT and the choice vector is a 129th product, C = reduce(sb) (c128, :105-106), one byte per OT (:108-115).
// sb has only `parts` set bits, so that product is a sum of `parts` cyclic shifts of a' = a mod (x^P - 1) -- no
// transform:  C[k] = [k in S0'] ^ XOR_{s in S1} a'[(k - s) mod P]
//   S0' = {S mod P : S < nb128} (the b0 half of the row, folded), S1 = {(S - nb128) mod P : nb128 <= S < 2 nb128}
// (row segments split at multiples of Block n_blocks, quasi_cyclic_encode.rs:267; points in the last cols - scaler n_blocks
// Blocks are outside every segment; with scaler > 2 segment s >= 1 contributes shifts of a_s').
// pts[i] = pos | (s - 1) << 56 | kind << 63 (kind 1: an S0' position), ~0 = dropped. One thread per 32 OTs.
#define SC_DROPPED (~0ull)
__device__ __forceinline__ uint32_t cyc32(const uint32_t *__restrict__ a, uint64_t P, uint64_t start) {
    // bits a'[(start + i) mod P], i < 32; a' has zero bits from P on and two padding words
    const uint64_t w0 = start >> 5;
    const uint32_t sh = (uint32_t)(start & 31);
    uint32_t v = a[w0] >> sh;
    if (sh) v |= a[w0 + 1] << (32 - sh);
    const uint64_t rem = P - start;  // >= 1
    if (rem < 32) v |= a[0] << rem;  // P > 32: at most one wrap
    return v;
}
__global__ void __launch_bounds__(256)
    k_sparse_choice(const uint32_t *__restrict__ a2, uint64_t a2_stride, uint64_t P, const uint64_t *__restrict__ pts,
                    uint32_t npts, uint64_t n, uint8_t *__restrict__ C) {
    __shared__ uint64_t sp[256];
    const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t k0 = 32 * w;
    uint32_t v = 0;
    // the points pass through LDS in chunks of 256 (`parts` is 120 or 128 for scaler 2 at 128-bit security, but any
    // geometry sot_configure accepts works: quasi_cyclic_encode.rs:104-117 has no limit)
    for (uint32_t c0 = 0; c0 < npts; c0 += 256) {
        const uint32_t cn = npts - c0 < 256 ? npts - c0 : 256;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < cn; i += blockDim.x) sp[i] = pts[c0 + i];
        __syncthreads();
        if (k0 < n)
            for (uint32_t i = 0; i < cn; i++) {
                const uint64_t e = sp[i];
                if (e == SC_DROPPED) continue;
                const uint64_t pos = e & ((1ull << 56) - 1);
                if (e >> 63) {
                    if ((pos >> 5) == w) v ^= 1u << (pos & 31);
                } else {  // bits 56..62: s - 1 of the row segment the point lies in, i.e. which public polynomial a_s
                    v ^= cyc32(a2 + ((e >> 56) & 127u) * a2_stride, P, k0 >= pos ? k0 - pos : k0 + P - pos);
                }
            }
    }
    if (k0 >= n) return;
    if (k0 + 32 <= n) {
        uint32_t o[8];
#pragma unroll
        for (int q = 0; q < 8; q++)
            o[q] = ((v >> (4 * q)) & 1u) | (((v >> (4 * q + 1)) & 1u) << 8) | (((v >> (4 * q + 2)) & 1u) << 16) |
                   (((v >> (4 * q + 3)) & 1u) << 24);
        uint4 *d = (uint4 *)(C + k0);
        d[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
        for (uint64_t k = k0; k < n; k++) C[k] = (uint8_t)((v >> (k - k0)) & 1u);
    }
}

// ------------------------------------------------------------------------------------------------
// K1: Blocks -> row words. Z[w * ROWS + row] bit p = bit `row` of V1'[32 w + p],
// V1'[c] = V1[c] ^ V1[c + P] (reduction mod x^P - 1 of the 2*n_blocks*128-bit row), c < lo_bits; zero above.
// One workgroup per 64 words (2048 Blocks in, 32 KiB out); both sides coalesced through LDS.
// One workgroup (128 threads) per 32 words (1024 Blocks in, 16 KiB out); both sides coalesced through one LDS
// buffer used in place (the kernel is latency bound: small workgroups, many per CU).
#define LD_W 32
#define LD_T 128
// FULL: all 128 rows live and whole-Block leaves (every context but the sharded ones of 4 or more ranks): rows_active and
// lw are compile-time constants, the kernel is the one-GPU kernel unchanged
template <bool FULL>
__global__ void __launch_bounds__(LD_T)
    k_qc_load(const blk_t *__restrict__ V, const uint32_t *__restrict__ sb, uint32_t sbpos, uint64_t v1_len, uint64_t P,
              uint64_t lo_bits, uint64_t nw_half, uint32_t *__restrict__ Z, uint32_t rows_active_, uint32_t lw_) {
    const uint32_t rows_active = FULL ? ROWS : rows_active_, lw = FULL ? 4u : lw_;
    __shared__ uint32_t tio[LD_W * 132];  // in: [w][p][q] words (stride 132); out: [w][row] (stride 129)
    const uint64_t w0 = (uint64_t)blockIdx.x * LD_W;
    for (uint32_t i = threadIdx.x; i < LD_W * 32; i += LD_T) {
        const uint64_t c = w0 * 32 + i;
        blk_t v = make_uint4(0, 0, 0, 0);
        if (c < lo_bits) {
            v = ld_leaf(V, v1_len + c, sb, sbpos, lw);  // V1 = V + nb128 and v1_len == nb128
            if (c + P < v1_len) {
                blk_t u = ld_leaf(V, v1_len + c + P, sb, sbpos, lw);
                v.x ^= u.x, v.y ^= u.y, v.z ^= u.z, v.w ^= u.w;
            }
        }
        uint32_t *d = tio + (i >> 5) * 132 + (i & 31) * 4;
        d[0] = v.x, d[1] = v.y, d[2] = v.z, d[3] = v.w;
    }
    __syncthreads();
    {
        // rows >= rows_active hold no data (sharded contexts, qc_set_rows_active) and are never read: their quarter of the
        // Block words is neither transposed nor written
        const uint32_t wl = threadIdx.x >> 2, q = threadIdx.x & 3;
        const bool live = 32 * q < rows_active;
        uint32_t a[32];
        if (live) {
#pragma unroll
            for (int p = 0; p < 32; p++) a[p] = tio[wl * 132 + p * 4 + q];
            transpose32(a);
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int b = 0; b < 32; b++) tio[wl * 129 + 32 * q + b] = a[b];
        }
    }
    __syncthreads();
    const uint32_t rows_w = rows_active > 96 ? 128 : rows_active > 64 ? 96 : rows_active > 32 ? 64 : 32;  // whole quarters
    for (uint32_t i = threadIdx.x; i < LD_W * ROWS; i += LD_T) {
        const uint64_t w = w0 + (i >> 7);
        if (w < nw_half && (i & 127) < rows_w) Z[w * ROWS + (i & 127)] = tio[(i >> 7) * 129 + (i & 127)];
    }
}

// public polynomial: row 0 from a packed bit vector (other rows zeroed by the caller)
__global__ void k_a_load(const uint32_t *__restrict__ abits, uint64_t nbits, uint64_t nw_half, uint32_t *__restrict__ Z) {
    uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nw_half) return;
    uint32_t v = 0;
    if (32 * w < nbits) {
        v = abits[w];
        if (nbits - 32 * w < 32) v &= (1u << (nbits - 32 * w)) - 1u;
    }
    Z[w * ROWS] = v;
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
// K2u: the 32 phases local to a unit of 2^16 coefficients, tile = 2048 words x 16 rows (4 row quads), in three parts:
//   P1  the 8 Taylor phases at x^256 + x (chunk 2^9 .. 2^16, shift 1 .. 128 bits): fold / shift steps in LDS;
//   P2  the 12 phases that move whole 256-bit sub-units (conversion of the 256 sub-unit indices): a host-built list
//       of (dst, src) sub-unit moves in hazard-free steps, 32 lanes per move;
//   P3  the 12 phases inside a 256-bit sub-unit: in registers, thread = (sub-unit, row quad), 8 x u4, no barriers.
// P2 and P3 commute. Forward = P1, P2, P3 (written to HBM straight from the registers); inverse = the reverse.
// LDS index of (word, quad) = 36 * (word / 8) + 4 * (word % 8) + quad: 64 B of padding per sub-unit keeps the
// 128-bit accesses of a quarter wave on distinct banks when threads own sub-units.
#define UP_MAX_STEPS 64
#define CU_MAX_MOVES 1536  // 12 sub-unit phases x at most 128 moves
// RQ = row quads per tile (4: 16 rows, 128 KiB + the move list, one workgroup of 1024 threads per CU; 2: 8 rows, 64 KiB,
// workgroups of 512 threads, TWO per CU, so that one's barrier-separated LDS phases overlap the other's)
#ifndef QC_CU_RQ
#define QC_CU_RQ 2
#endif
template <int RQ>
__device__ __forceinline__ uint32_t cu_idx(uint32_t word) { return (word + (word >> 3)) * RQ; }

template <int RQ>
__device__ __forceinline__ void cu_fold(u4 *lds, uint32_t lchunk, uint32_t lpw) {
    constexpr uint32_t T = 256 * RQ;
    const uint32_t chunk_w = 1u << (lchunk - 5), half_w = chunk_w >> 1;
    const uint32_t nchunks = 2048u >> (lchunk - 5);
    if (lpw >= 5) {
        const uint32_t pw_w = 1u << (lpw - 5);
        const uint32_t items = nchunks * pw_w * RQ;
        for (uint32_t it = threadIdx.x; it < items; it += T) {
            uint32_t r = it % RQ, wi = it / RQ;
            uint32_t c = wi >> (lpw - 5), d = wi & (pw_w - 1);
            uint32_t base = c * chunk_w + half_w;
            lds[cu_idx<RQ>(base + d) + r] ^= lds[cu_idx<RQ>(base + half_w - pw_w + d) + r];
        }
    } else {
        const uint32_t pw = 1u << lpw;
        const uint32_t items = nchunks * RQ;
        for (uint32_t it = threadIdx.x; it < items; it += T) {
            uint32_t r = it % RQ, c = it / RQ;
            uint32_t base = c * chunk_w + half_w;
            lds[cu_idx<RQ>(base) + r] ^= lds[cu_idx<RQ>(base + half_w - 1) + r] >> (32 - pw);
        }
    }
}

template <int RQ>
__device__ __forceinline__ void cu_shift(u4 *lds, uint32_t lchunk, uint32_t lpw) {
    constexpr uint32_t T = 256 * RQ;
    const uint32_t chunk_w = 1u << (lchunk - 5), half_w = chunk_w >> 1;
    const uint32_t items = 1024 * RQ;
    if (lpw >= 5) {
        const uint32_t pw_w = 1u << (lpw - 5);
        for (uint32_t it = threadIdx.x; it < items; it += T) {
            uint32_t r = it % RQ, wi = it / RQ;
            uint32_t c = wi >> (lchunk - 6), d = wi & (half_w - 1);
            if (d >= pw_w) {
                uint32_t base = c * chunk_w;
                lds[cu_idx<RQ>(base + d) + r] ^= lds[cu_idx<RQ>(base + half_w + d - pw_w) + r];
            }
        }
    } else {
        const uint32_t pw = 1u << lpw;
        for (uint32_t it = threadIdx.x; it < items; it += T) {
            uint32_t r = it % RQ, wi = it / RQ;
            uint32_t c = wi >> (lchunk - 6), d = wi & (half_w - 1);
            uint32_t base = c * chunk_w;
            u4 h = lds[cu_idx<RQ>(base + half_w + d) + r] << pw;
            if (d) h |= lds[cu_idx<RQ>(base + half_w + d - 1) + r] >> (32 - pw);
            lds[cu_idx<RQ>(base + d) + r] ^= h;
        }
    }
}

// masks of one phase for word wd of a 256-bit sub-unit (gf2fft_model.apply_phase)
constexpr uint32_t rp_m1(int lc, int lp, int wd) {
    uint32_t m = 0;
    const int chunk = 1 << lc, pw = 1 << lp, half = chunk >> 1;
    for (int b = 0; b < 32; b++) {
        const int l = (32 * wd + b) % chunk;
        if (l >= half ? (l - half < pw) : (l >= pw)) m |= 1u << b;
    }
    return m;
}
constexpr uint32_t rp_m2(int lc, int lp, int wd) {
    uint32_t m = 0;
    const int chunk = 1 << lc, pw = 1 << lp, half = chunk >> 1;
    for (int b = 0; b < 32; b++) {
        const int l = (32 * wd + b) % chunk;
        if (l < half && l >= pw && l < 2 * pw) m |= 1u << b;
    }
    return m;
}
// T = u4 (four rows at once) or uint32_t (one row: a quarter of the registers)
template <int BIT, class T>
__device__ __forceinline__ T rp_fetch(const T w[8]) {  // bits BIT .. BIT+31 of the 256-bit value, zero beyond
    constexpr int ws = BIT / 32, sh = BIT % 32;
    if (ws >= 8) return T{};
    T lo = w[ws];
    if (sh == 0) return lo;
    T r = lo >> sh;
    if (ws + 1 < 8) r |= w[ws + 1] << (32 - sh);
    return r;
}
template <int LC, int LP, bool INV, int WD, class T>
__device__ __forceinline__ T rp_word(const T w[8]) {
    constexpr int D = (1 << (LC - 1)) - (1 << LP);
    constexpr uint32_t m1 = rp_m1(LC, LP, WD), m2 = INV ? 0u : rp_m2(LC, LP, WD);
    T o = w[WD];
    if (m1) o ^= rp_fetch<32 * WD + D, T>(w) & m1;
    if (m2) o ^= rp_fetch<32 * WD + 2 * D, T>(w) & m2;
    return o;
}
template <int LC, int LP, bool INV, class T>
__device__ __forceinline__ void reg_phase(T w[8]) {
    T o[8];
    o[0] = rp_word<LC, LP, INV, 0, T>(w), o[1] = rp_word<LC, LP, INV, 1, T>(w), o[2] = rp_word<LC, LP, INV, 2, T>(w);
    o[3] = rp_word<LC, LP, INV, 3, T>(w), o[4] = rp_word<LC, LP, INV, 4, T>(w), o[5] = rp_word<LC, LP, INV, 5, T>(w);
    o[6] = rp_word<LC, LP, INV, 6, T>(w), o[7] = rp_word<LC, LP, INV, 7, T>(w);
#pragma unroll
    for (int j = 0; j < 8; j++) w[j] = o[j];
}
// conversion_phases(8): (8,3) (7,2) (6,1) (5,0) (8,5) (7,4) (8,6) (6,4) (4,1) (3,0) (4,2) (2,0)
template <bool INV, class T>
__device__ __forceinline__ void reg_phases(T w[8]) {
    if (!INV) {
        reg_phase<8, 3, false, T>(w), reg_phase<7, 2, false, T>(w), reg_phase<6, 1, false, T>(w);
        reg_phase<5, 0, false, T>(w), reg_phase<8, 5, false, T>(w), reg_phase<7, 4, false, T>(w);
        reg_phase<8, 6, false, T>(w), reg_phase<6, 4, false, T>(w), reg_phase<4, 1, false, T>(w);
        reg_phase<3, 0, false, T>(w), reg_phase<4, 2, false, T>(w), reg_phase<2, 0, false, T>(w);
    } else {
        reg_phase<2, 0, true, T>(w), reg_phase<4, 2, true, T>(w), reg_phase<3, 0, true, T>(w);
        reg_phase<4, 1, true, T>(w), reg_phase<6, 4, true, T>(w), reg_phase<8, 6, true, T>(w);
        reg_phase<7, 4, true, T>(w), reg_phase<8, 5, true, T>(w), reg_phase<5, 0, true, T>(w);
        reg_phase<6, 1, true, T>(w), reg_phase<7, 2, true, T>(w), reg_phase<8, 3, true, T>(w);
    }
}

// XCD-aware tile order for the kernels whose tiles are row slices (16 or 32 of the 128 rows: 64-B or 128-B pieces of every
// 512-B word). Workgroup v runs on XCD v % 8 and each XCD has its own L2; with the slice index as the fastest tile
// coordinate the nrg slices of a word land on nrg different XCDs, every L2 fetches its own sector of each line and HBM sees
// 64-B requests scattered in time (3.4 TB/s read, 3.7 read+write in tools/micro/gather_copy.hip). Sending the nrg
// slices of the same words to nrg workgroups of ONE XCD, adjacent in launch order, makes them meet in that L2
// (6.1 / 5.5 TB/s in the same test). v -> tile index (other coordinates) * nrg + slice; a bijection when ntiles is a
// multiple of 8 nrg (identity otherwise: small transforms). nrg is a power of two.
__device__ __forceinline__ uint32_t xcd_slice_order(uint32_t v, uint32_t nrg, uint32_t ntiles) {
    if (ntiles % (8 * nrg)) return v;
    const uint32_t x = v & 7u, q = v >> 3;
    return ((q / nrg) * 8 + x) * nrg + (q & (nrg - 1u));
}

struct move_list {
    uint32_t nsteps;
    uint32_t step_end[UP_MAX_STEPS];
    const uint32_t *list;  // PH_ENTRY words (sub-unit indices): one step per phase
};
// INV: thread owns sub-unit su (8 consecutive words) x row quad; forward: words su, su + 256, ...
// (eight named registers, not an array: as a loop-carried array the prefetched tile ends up in scratch memory)
// a 16-byte load as values (a plain u4 struct copy is a memcpy that can pin its destination in scratch memory)
__device__ __forceinline__ u4 ld_u4(const uint32_t *p) {
    const uint4 t = *(const uint4 *)p;
    return {t.x, t.y, t.z, t.w};
}
struct cu_regs {
    u4 a, b, c, d, e, f, g, h;
};
template <bool INV>
__device__ __forceinline__ cu_regs cu_load_tile(const uint32_t *g, uint32_t su, uint32_t quad) {  // g: the tile's first row
    const uint32_t *p = g + (uint64_t)(INV ? 8 * su : su) * ROWS + 4 * quad;
    constexpr uint64_t st = (INV ? 1 : 256) * ROWS;
    cu_regs r;
    r.a = ld_u4(p), r.b = ld_u4(p + st), r.c = ld_u4(p + 2 * st), r.d = ld_u4(p + 3 * st);
    r.e = ld_u4(p + 4 * st), r.f = ld_u4(p + 5 * st), r.g = ld_u4(p + 6 * st), r.h = ld_u4(p + 7 * st);
    return r;
}

// Persistent: workgroups walk tiles blockIdx.x, blockIdx.x + gridDim.x, ...; the next tile's 32 registers per thread are
// loaded while the LDS phases of the current one run (with RQ = 4 the tile fills the CU's LDS and this is the only
// overlap there is; with RQ = 2 a second workgroup on the CU overlaps its phases with this one's barriers as well).
template <bool INV, int RQ>
__global__ void __launch_bounds__(256 * RQ) __attribute__((amdgpu_waves_per_eu(4, 4)))
k_conv_unit(uint32_t *__restrict__ D, move_list mv, uint32_t ntiles, uint32_t nact) {
    constexpr uint32_t T = 256 * RQ;      // threads
    constexpr uint32_t SU = 9 * RQ;       // u4 per sub-unit in LDS: 8 words x RQ quads + one word of padding
    constexpr uint32_t LANES = 8 * RQ;    // u4 of a sub-unit = lanes per move in P2
    __shared__ u4 lds[(2048 + 256) * RQ];
    __shared__ uint32_t mvl[CU_MAX_MOVES];  // the move list, staged once (a global load in P2 would drain the prefetch)
    const uint32_t su = threadIdx.x / RQ, quad = threadIdx.x % RQ;  // owner mapping of part P3
    // ntiles = units x nact: the first nact (32 / RQ, or fewer in a sharded context: qc_set_rows_active) slices of 4 RQ rows
    auto gptr = [&](uint32_t v) {
        const uint32_t t = xcd_slice_order(v, nact, ntiles);
        return D + (uint64_t)(t / nact) * 2048 * ROWS + (t & (nact - 1u)) * (4 * RQ);
    };
    // (a fixed list of tiles per workgroup: drawing them from a counter as k_mid does gains 3 % here and costs 18 % more HBM
    // traffic -- the slices of a line no longer meet in their XCD's L2 at the same time: profiles/r3_notes.md)
    uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    for (uint32_t i = threadIdx.x; i < mv.step_end[mv.nsteps - 1]; i += T) mvl[i] = mv.list[i];
    cu_regs nx = cu_load_tile<INV>(gptr(t), su, quad);
#pragma unroll 1
    for (; t < ntiles; t += gridDim.x) {
        uint32_t *g = gptr(t);
        {
            u4 w[8] = {nx.a, nx.b, nx.c, nx.d, nx.e, nx.f, nx.g, nx.h};
            if (!INV) {
#pragma unroll
                for (int j = 0; j < 8; j++) lds[cu_idx<RQ>(su + 256 * j) + quad] = w[j];
            } else {
                reg_phases<true, u4>(w);
#pragma unroll
                for (int j = 0; j < 8; j++) lds[SU * su + RQ * j + quad] = w[j];
            }
        }
        __syncthreads();
        {  // next tile: in flight during the LDS phases (the last iteration re-reads its own tile: unconditional
           // loads keep w[] in registers)
            nx = cu_load_tile<INV>(gptr(t + gridDim.x < ntiles ? t + gridDim.x : t), su, quad);
        }
        // P2 forward comes after P1, inverse before it
        for (int part = 0; part < 2; part++) {
            const bool do_p1 = INV ? part == 1 : part == 0;
            if (do_p1) {
                for (int q = 0; q < 8; q++) {
                    const uint32_t lpw = INV ? q : 7 - q, lchunk = lpw + 9;
                    if (!INV) {
                        cu_fold<RQ>(lds, lchunk, lpw);
                        __syncthreads();
                        cu_shift<RQ>(lds, lchunk, lpw);
                    } else {
                        cu_shift<RQ>(lds, lchunk, lpw);
                        __syncthreads();
                        cu_fold<RQ>(lds, lchunk, lpw);
                    }
                    __syncthreads();
                }
            } else {
                const uint32_t x = threadIdx.x % LANES, grp = threadIdx.x / LANES;
                uint32_t start = 0;
                for (uint32_t st = 0; st < mv.nsteps; st++) {
                    const uint32_t end = mv.step_end[st];
                    for (uint32_t it = start + grp; it < end; it += T / LANES) {  // entries: see PH_ENTRY
                        const uint32_t e = mvl[it];
                        const uint32_t a = SU * (e & 1023u) + x, c = SU * ((e >> 10) & 1023u) + x;
                        u4 h = lds[a];
                        if (e >> 30) {
                            const u4 gg = lds[SU * ((e >> 20) & 1023u) + x];
                            if (!INV) h ^= gg;
                            lds[a] = INV ? h ^ gg : h;
                        }
                        lds[c] ^= h;
                    }
                    start = end;
                    __syncthreads();
                }
            }
        }
        if (!INV) {
            u4 o[8];
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = lds[SU * su + RQ * j + quad];
            // one row at a time: the prefetched tile already holds 32 registers
            {
                uint32_t x[8];
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = o[j].x;
                reg_phases<false, uint32_t>(x);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j].x = x[j];
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = o[j].y;
                reg_phases<false, uint32_t>(x);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j].y = x[j];
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = o[j].z;
                reg_phases<false, uint32_t>(x);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j].z = x[j];
#pragma unroll
                for (int j = 0; j < 8; j++) x[j] = o[j].w;
                reg_phases<false, uint32_t>(x);
#pragma unroll
                for (int j = 0; j < 8; j++) o[j].w = x[j];
            }
#pragma unroll
            for (int j = 0; j < 8; j++) *(u4 *)(g + (uint64_t)(8 * su + j) * ROWS + 4 * quad) = o[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++) *(u4 *)(g + (uint64_t)(su + 256 * j) * ROWS + 4 * quad) = lds[cu_idx<RQ>(su + 256 * j) + quad];
        }
        __syncthreads();  // the tile's LDS is free again
    }
}

// ------------------------------------------------------------------------------------------------
// K2c: the Taylor-expansion phases at y = x^(2^16) + x whose chunk spans several units but whose shift pw is
// smaller than a unit (gf2fft_model.conversion_phases: chunk = 2^(pb+17), pw = 2^pb), fused in LDS, out of place.
// Tile = the NU units of one independent group x NC columns (words at the same offsets of every unit) x 4 RQ rows.
// The first HW columns are a halo: the words that linearly precede the payload columns (offsets o0-HW .. o0-1, or,
// for the tile at o0 = 0 ("wrap"), the LAST HW words of each unit, which precede offset 0 of the next unit).
// Every phase only pulls data from lower linear offsets (shift: lo[e] ^= H[e - pw]) or from the end of a unit
// into the start of one (fold: H[j] ^= H[j + half - pw], j < pw), so with HW >= the sum of the shifts the payload
// is exact; the halo decays from its low end and is never written back.
//   group A: unit 2^21 bits, pw = 2^5 .. 2^(R-1) bits (whole words)      group B: unit 2^16 bits, pw = 1 .. 16 bits
struct taylor_desc {
    uint32_t lus;  // log2(unit bits)
    uint32_t lnu;  // log2(units per group)
    uint32_t lnc;  // log2(columns per tile, halo included)
    uint32_t hw;   // halo columns
    uint32_t lrq;  // log2(row quads per tile)
    uint32_t lg;   // log2(column stride in words): every shift of the tile's phases is a multiple of 2^lg words, so
                   // the words with equal index mod 2^lg form independent problems; a tile takes one residue g
    uint32_t nph;
    uint32_t nact;  // row groups (of 4 << lrq rows) that hold data: 32 >> lrq, or fewer in a sharded context; a power of two
    uint8_t lchunk[8], lpw[8];
};

__device__ __forceinline__ u4 ty_fetch(const u4 *lds, const taylor_desc &t, bool wrap, uint32_t us, uint32_t d,
                                       uint32_t col, int32_t cs, uint32_t r) {
    const u4 zero = {0, 0, 0, 0};
    if (cs < 0) return zero;
    if (wrap && col >= t.hw && (uint32_t)cs < t.hw) {  // crossed offset 0: end of the preceding unit of H
        if (d == 0) return zero;
        us -= 1;
    }
    return lds[(((us << t.lnc) + (uint32_t)cs) << t.lrq) + r];
}

__device__ __forceinline__ void ty_shift(u4 *lds, const taylor_desc &t, bool wrap, uint32_t lchunk, uint32_t lpw) {
    const uint32_t lhalf = lchunk - 1 - t.lus;  // log2(units per half chunk)
    const uint32_t half_u = 1u << lhalf;
    const uint32_t sw = ((1u << lpw) >> 5) >> t.lg, sb = (1u << lpw) & 31u;
    const uint32_t items = 1u << (t.lnu - 1 + t.lnc + t.lrq);
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        const uint32_t r = it & ((1u << t.lrq) - 1u);
        const uint32_t col = (it >> t.lrq) & ((1u << t.lnc) - 1u);
        const uint32_t ui = it >> (t.lrq + t.lnc);
        const uint32_t c = ui >> lhalf, d = ui & (half_u - 1);
        const uint32_t ud = (c << (lhalf + 1)) + d, us = ud + half_u;
        u4 h = ty_fetch(lds, t, wrap, us, d, col, (int32_t)col - (int32_t)sw, r);
        if (sb) h = (h << sb) | (ty_fetch(lds, t, wrap, us, d, col, (int32_t)col - (int32_t)sw - 1, r) >> (32 - sb));
        lds[(((ud << t.lnc) + col) << t.lrq) + r] ^= h;
    }
}

__device__ __forceinline__ void ty_fold(u4 *lds, const taylor_desc &t, bool wrap, uint32_t lchunk, uint32_t lpw) {
    if (!wrap) return;
    const uint32_t lhalf = lchunk - 1 - t.lus;
    const uint32_t half_u = 1u << lhalf;
    const uint32_t nchunks = 1u << (t.lnu - lhalf - 1);
    const uint32_t sw = ((1u << lpw) >> 5) >> t.lg, sb = (1u << lpw) & 31u;
    const uint32_t nx = sb ? 1u : sw;
    const uint32_t items = (nchunks * nx) << t.lrq;
    for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
        const uint32_t r = it & ((1u << t.lrq) - 1u);
        const uint32_t x = (it >> t.lrq) % nx, c = (it >> t.lrq) / nx;
        const uint32_t uh0 = (c << (lhalf + 1)) + half_u, uhl = uh0 + half_u - 1;
        u4 v = lds[(((uhl << t.lnc) + t.hw - nx + x) << t.lrq) + r];
        if (sb) v = v >> (32 - sb);
        lds[(((uh0 << t.lnc) + t.hw + x) << t.lrq) + r] ^= v;
    }
}

// element `it` (of the 8192 of a tile) of tile `b`, zero outside the unit
__device__ __forceinline__ u4 ty_ld(const uint32_t *__restrict__ Din, const taylor_desc &t, uint32_t ncoltiles, uint32_t b,
                                    uint32_t it) {
    const uint32_t UW = 1u << (t.lus - 5 - t.lg);  // columns per unit
    const uint32_t NC = 1u << t.lnc, W = NC - t.hw;
    const uint32_t nrg = t.nact;  // row groups
    const uint32_t rg = b % nrg, g = (b / nrg) & ((1u << t.lg) - 1u), bc = (b / nrg) >> t.lg;
    const uint32_t cb = bc % ncoltiles, grp = bc / ncoltiles;
    const uint32_t o0 = cb * W;
    const bool wrap = o0 == 0;
    const uint64_t gbase = (((uint64_t)grp << (t.lnu + t.lus - 5)) + g) * ROWS + (rg << (t.lrq + 2));
    const uint32_t r = it & ((1u << t.lrq) - 1u);
    const uint32_t col = (it >> t.lrq) & (NC - 1);
    const uint32_t u = it >> (t.lrq + t.lnc);
    const uint32_t o = col >= t.hw ? o0 + col - t.hw : (wrap ? UW - t.hw + col : o0 - t.hw + col);
    u4 v = {0, 0, 0, 0};
    if (o < UW) v = ld_u4(Din + gbase + ((((uint64_t)u * UW + o)) << t.lg) * ROWS + 4 * r);
    return v;
}
__device__ __forceinline__ cu_regs ty_fetch_tile(const uint32_t *__restrict__ Din, const taylor_desc &t, uint32_t ncoltiles,
                                                 uint32_t b) {
    cu_regs x;
    const uint32_t i = threadIdx.x;
    x.a = ty_ld(Din, t, ncoltiles, b, i), x.b = ty_ld(Din, t, ncoltiles, b, i + CT_THREADS);
    x.c = ty_ld(Din, t, ncoltiles, b, i + 2 * CT_THREADS), x.d = ty_ld(Din, t, ncoltiles, b, i + 3 * CT_THREADS);
    x.e = ty_ld(Din, t, ncoltiles, b, i + 4 * CT_THREADS), x.f = ty_ld(Din, t, ncoltiles, b, i + 5 * CT_THREADS);
    x.g = ty_ld(Din, t, ncoltiles, b, i + 6 * CT_THREADS), x.h = ty_ld(Din, t, ncoltiles, b, i + 7 * CT_THREADS);
    return x;
}

// Persistent (one workgroup per CU, tiles b = blockIdx.x, + gridDim.x, ...; a tile is always 8192 elements = 8 per
// thread): the next tile is fetched into registers while the phases of the current one run in LDS.
__global__ void __launch_bounds__(CT_THREADS)
    k_conv_taylor(const uint32_t *__restrict__ Din, uint32_t *__restrict__ Dout, taylor_desc t, uint32_t ncoltiles,
                  int inverse, uint32_t ntiles) {
    __shared__ u4 lds[8192];
    const uint32_t UW = 1u << (t.lus - 5 - t.lg);  // columns per unit
    const uint32_t NC = 1u << t.lnc, W = NC - t.hw;
    const uint32_t nrg = t.nact;  // row groups
    const uint32_t items = 1u << (t.lnu + t.lnc + t.lrq);  // == 8192 (make_taylor)
    uint32_t vb = blockIdx.x;
    if (vb >= ntiles) return;
    cu_regs nx = ty_fetch_tile(Din, t, ncoltiles, xcd_slice_order(vb, nrg, ntiles));
#pragma unroll 1
    for (; vb < ntiles; vb += gridDim.x) {
        const uint32_t b = xcd_slice_order(vb, nrg, ntiles);
        const uint32_t rg = b % nrg, g = (b / nrg) & ((1u << t.lg) - 1u), bc = (b / nrg) >> t.lg;
        const uint32_t cb = bc % ncoltiles;
        const uint32_t grp = bc / ncoltiles;
        const uint32_t o0 = cb * W;
        const bool wrap = o0 == 0;
        const uint64_t gbase = (((uint64_t)grp << (t.lnu + t.lus - 5)) + g) * ROWS + (rg << (t.lrq + 2));
        {
            const uint32_t i = threadIdx.x;
            lds[i] = nx.a, lds[i + CT_THREADS] = nx.b, lds[i + 2 * CT_THREADS] = nx.c, lds[i + 3 * CT_THREADS] = nx.d;
            lds[i + 4 * CT_THREADS] = nx.e, lds[i + 5 * CT_THREADS] = nx.f, lds[i + 6 * CT_THREADS] = nx.g;
            lds[i + 7 * CT_THREADS] = nx.h;
        }
        __syncthreads();
        nx = ty_fetch_tile(Din, t, ncoltiles, xcd_slice_order(vb + gridDim.x < ntiles ? vb + gridDim.x : vb, nrg, ntiles));
        for (uint32_t q = 0; q < t.nph; q++) {
            const uint32_t i = inverse ? t.nph - 1 - q : q;
            const uint32_t lchunk = t.lchunk[i], lpw = t.lpw[i];
            if (!inverse) {
                ty_fold(lds, t, wrap, lchunk, lpw);
                __syncthreads();
                ty_shift(lds, t, wrap, lchunk, lpw);
            } else {
                ty_shift(lds, t, wrap, lchunk, lpw);
                __syncthreads();
                ty_fold(lds, t, wrap, lchunk, lpw);
            }
            __syncthreads();
        }
        for (uint32_t it = threadIdx.x; it < items; it += CT_THREADS) {
            const uint32_t r = it & ((1u << t.lrq) - 1u);
            const uint32_t col = (it >> t.lrq) & (NC - 1);
            const uint32_t u = it >> (t.lrq + t.lnc);
            if (col < t.hw) continue;
            const uint32_t o = o0 + col - t.hw;
            if (o < UW) *(u4 *)(Dout + gbase + (((uint64_t)u * UW + o) << t.lg) * ROWS + 4 * r) = lds[it];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K2w: the whole-unit phases (chunk and pw multiples of a 2^16 unit) when the top butterfly pass cannot host them
// (m >= 26: more unit-index bits than its tile has). They only XOR granules that sit at the same offset of different
// units, so a tile = ALL 2^nub units x one word offset x 16 .. 128 rows (64 B per unit at nub = 11, 128 KiB) takes every
// phase in one HBM round trip, in place; before, each phase was a pass of k_conv_phase over the whole array (85 ms
// of a 155 ms step at n = 2^26). Persistent with next-tile prefetch like k_conv_unit.
struct wu_phases {
    uint32_t nub;  // log2(units)
    uint32_t nact;  // row groups (of 4 << lq rows) that hold data: 32 >> lq, or fewer in a sharded context; a power of two
    uint32_t nph;
    uint8_t lchunk[32], lpw[32];  // log2 of chunk and pw in units; half = chunk / 2
};
// tile = all 2^nub units x one word offset x 2^lq row quads, lq = 13 - nub (4 quads = 64 B per unit at nub = 11, the
// whole 512-B granule at nub = 8): always 8192 u4 = 128 KiB, [unit][quad]
__device__ __forceinline__ cu_regs wu_fetch(const uint32_t *__restrict__ D, uint32_t lq, uint32_t nact, uint32_t vtile) {
    const uint32_t tile = xcd_slice_order(vtile, nact, 2048u * nact);
    const uint32_t o = tile / nact, rg = tile & (nact - 1u);
    const uint32_t *g = D + (uint64_t)o * ROWS + (rg << (lq + 2));
    auto ld = [&](uint32_t it) { return ld_u4(g + (uint64_t)(it >> lq) * 2048 * ROWS + 4 * (it & ((1u << lq) - 1u))); };
    cu_regs x;
    const uint32_t i = threadIdx.x;
    x.a = ld(i), x.b = ld(i + CT_THREADS), x.c = ld(i + 2 * CT_THREADS), x.d = ld(i + 3 * CT_THREADS);
    x.e = ld(i + 4 * CT_THREADS), x.f = ld(i + 5 * CT_THREADS), x.g = ld(i + 6 * CT_THREADS), x.h = ld(i + 7 * CT_THREADS);
    return x;
}
__global__ void __launch_bounds__(CT_THREADS) k_conv_wholeunits(uint32_t *__restrict__ D, wu_phases ph, int inverse) {
    __shared__ u4 lds[8192];  // [unit][quad]
    const uint32_t lq = 13 - ph.nub, qm = (1u << lq) - 1u;  // 8 <= nub <= 11 (run_conversion)
    const uint32_t nact = ph.nact;
    const uint32_t ntiles = 2048u * nact;
    uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    cu_regs nx = wu_fetch(D, lq, nact, t);
#pragma unroll 1
    for (; t < ntiles; t += gridDim.x) {
        {
            const uint32_t i = threadIdx.x;
            lds[i] = nx.a, lds[i + CT_THREADS] = nx.b, lds[i + 2 * CT_THREADS] = nx.c, lds[i + 3 * CT_THREADS] = nx.d;
            lds[i + 4 * CT_THREADS] = nx.e, lds[i + 5 * CT_THREADS] = nx.f, lds[i + 6 * CT_THREADS] = nx.g;
            lds[i + 7 * CT_THREADS] = nx.h;
        }
        __syncthreads();
        nx = wu_fetch(D, lq, nact, t + gridDim.x < ntiles ? t + gridDim.x : t);
        for (uint32_t q = 0; q < ph.nph; q++) {
            const uint32_t pi = inverse ? ph.nph - 1 - q : q;
            const uint32_t lchunk = ph.lchunk[pi], lpw = ph.lpw[pi];
            const uint32_t half = 1u << (lchunk - 1), pw = 1u << lpw, Dd = half - pw;
            // fold (H[j] ^= H[j + D], j < pw) and shift (lo[pw + j] ^= H[j], j < D) of a phase in ONE hazard-free step:
            // the thread that owns index j does both (forward: fold first, inverse: shift first). H[j + D], j < pw,
            // lies in [half - pw, half) of H, which no thread writes; H[j] and lo[pw + j] are touched by j's owner
            // only. One barrier per phase instead of two.
            for (uint32_t it = threadIdx.x; it < 4096; it += CT_THREADS) {  // half of the units x 2^lq quads
                const uint32_t r = it & qm, w = it >> lq, c = w >> (lchunk - 1), j = w & (half - 1);
                if (j < Dd) {
                    const uint32_t base = c << lchunk;
                    u4 h = lds[((base + half + j) << lq) + r];
                    if (j < pw) {
                        const u4 g = lds[((base + half + j + Dd) << lq) + r];
                        if (!inverse) h ^= g;
                        lds[((base + half + j) << lq) + r] = inverse ? h ^ g : h;
                    }
                    lds[((base + pw + j) << lq) + r] ^= h;
                }
            }
            __syncthreads();
        }
        {
            const uint32_t tt = xcd_slice_order(t, nact, ntiles);
            const uint32_t o = tt / nact, rg = tt & (nact - 1u);
            uint32_t *g = D + (uint64_t)o * ROWS + (rg << (lq + 2));
            for (uint32_t it = threadIdx.x; it < 8192; it += CT_THREADS)
                *(u4 *)(g + (uint64_t)(it >> lq) * 2048 * ROWS + 4 * (it & qm)) = lds[it];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// twiddle matrices. TW[tbl][e][k] = row k (bit j set <=> output coordinate k takes input coordinate j) of
// "multiply by point(e << 8 tbl)", e < 256, tbl < 4; TWC[tbl][e] = that point. point() is GF(2)-linear, so the
// rows / the element of any index q are the XOR over tbl of the entries of q's bytes.
#define TW_TABLES 4
__global__ void k_build_tw(uint32_t *__restrict__ TW, uint32_t *__restrict__ TWC) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // (tbl, e, k)
    if (idx >= TW_TABLES * 256 * 32) return;
    uint32_t k = idx & 31, e = (idx >> 5) & 255, tbl = idx >> 13;
    uint32_t p = e << (8 * tbl);
    uint32_t c = 0;
    for (int i = 0; i < 32; i++)
        if ((p >> i) & 1) c ^= d_BETA[i];
    uint32_t row = 0;
    for (int t = 0; t < 32; t++)
        if ((c >> t) & 1) row ^= d_RTAB[t * 32 + k];
    TW[idx] = row;
    if (k == 0) TWC[tbl * 256 + e] = c;
}

// rows k0 .. k0+N-1 of the matrix of point(q); q and k0 are wave-uniform: everything stays in SGPRs
// NT = number of byte tables that q can touch (q < 2^(8 NT))
template <int N, int NT = 4>
__device__ __forceinline__ void tw_rows(const uint32_t *__restrict__ TW, uint32_t q, uint32_t k0, uint32_t rows[N]) {
    constexpr int AL = N >= 8 ? 32 : 4 * N;  // k0 is a multiple of N: lets the loads merge into s_load_dwordx8
    const uint32_t *tb = (const uint32_t *)__builtin_assume_aligned(TW, 128);
    const uint32_t *t0 = (const uint32_t *)__builtin_assume_aligned(tb + ((q & 255) << 5) + k0, AL);
    const uint32_t *t1 = (const uint32_t *)__builtin_assume_aligned(tb + ((256 + ((q >> 8) & 255)) << 5) + k0, AL);
    const uint32_t *t2 = (const uint32_t *)__builtin_assume_aligned(tb + ((512 + ((q >> 16) & 255)) << 5) + k0, AL);
    const uint32_t *t3 = (const uint32_t *)__builtin_assume_aligned(tb + ((768 + (q >> 24)) << 5) + k0, AL);
#pragma unroll
    for (int k = 0; k < N; k++) {
        rows[k] = t0[k];
        if (NT > 1) rows[k] ^= t1[k];
        if (NT > 2) rows[k] ^= t2[k];
        if (NT > 3) rows[k] ^= t3[k];
    }
}

// acc ^= M u for the N x N block M = rows[K0 + k] bits J0 + j (k, j < N) of a twiddle matrix, where M is the matrix of an
// element of GF(2^N): over the tower GF(2^N) = GF(2^(N/2))[t], t^2 = t + zeta, M = [[a, x], [b, a + b]] in blocks, so
//   acc_lo ^= a u_lo ^ x u_hi,   acc_hi ^= (a + b)(u_lo ^ u_hi) ^ a u_lo
// three half-size products instead of four, recursively down to 2 x 2 (measured: 3.44 -> 3.28 -> 3.22 ms per step for bases 8, 4, 2; scalar masks: a v_bitop3 with an SGPR operand
// costs 5 cycles, a plain v_xor 3, so trading 7 XANDs of 16 for 3-4 XORs pays twice over).
#ifndef KARA_BASE
#define KARA_BASE 2
#endif
template <int N, int K0, int J0>
__device__ __forceinline__ void kara_acc(uint32_t *acc, const uint32_t *u, const uint32_t *rows) {
    if constexpr (N <= KARA_BASE) {
#pragma unroll
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int k = 0; k < N; k++) acc[k] = XAND(acc[k], u[j], 0u - ((rows[K0 + k] >> (J0 + j)) & 1u));
        }
    } else {
        constexpr int H = N / 2;
        uint32_t t1[H], sm[H];
#pragma unroll
        for (int k = 0; k < H; k++) t1[k] = 0, sm[k] = u[k] ^ u[k + H];
        kara_acc<H, K0, J0>(t1, u, rows);
        kara_acc<H, K0, J0 + H>(acc, u + H, rows);
        kara_acc<H, K0 + H, J0 + H>(acc + H, sm, rows);
#pragma unroll
        for (int k = 0; k < H; k++) acc[k] ^= t1[k], acc[k + H] ^= t1[k];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ------------------------------------------------------------------------------------------------
// Tiles. A workgroup gathers 2^nb granules (x 64 rows) whose word indices differ in the nb bits pos[0..nb),
// runs some butterfly levels on them in LDS and scatters them back. Tile bit j <-> word-index bit pos[j]; the
// plane bits come first (tile bit t = plane-index bit t for every pruned level already passed), so the K = 2^np
// planes of a value are K consecutive LDS slots.
struct tile_desc {
    uint32_t m;      // log2 capacity
    uint32_t nb;     // tile bits (<= 9)
    uint32_t tmask;  // word-index bits that belong to the tile
    uint32_t pmw;    // word-index bits of the pruned levels
    uint32_t zero_bit;  // tile bit whose "1" half is known to be zero on input (forward transform of an operand); 32 = none
    uint32_t rhalves;   // 2: both 64-row halves of every tile are processed; 1: only rows 0..63 hold data (a sharded
                        // context of 4 or more ranks owns at most 64 bit rows: qc_set_rows_active), the grid is half as large
    uint32_t nlev;
    uint64_t posq;    // 5 bits per tile bit: its word-index bit (packed so that the kernels never index kernarg memory)
    uint32_t lev[8];  // r | tbit << 8 | np << 16 | pruned << 24
};

__device__ __forceinline__ uint32_t tile_base(const tile_desc &d, uint32_t o) {
    uint32_t w = 0;
    for (uint32_t b = 0; b + 5 < d.m; b++)
        if (!((d.tmask >> b) & 1u)) {
            w |= (o & 1u) << b;
            o >>= 1;
        }
    return w;
}
// tile index -> (gather-tile number, row half) under d.rhalves
__device__ __forceinline__ uint32_t tile_rh(const tile_desc &d, uint32_t idx) { return d.rhalves == 1 ? 0u : idx & 1u; }
__device__ __forceinline__ uint32_t tile_no(const tile_desc &d, uint32_t idx) { return d.rhalves == 1 ? idx : idx >> 1; }
__device__ __forceinline__ uint32_t tile_word(const tile_desc &d, uint32_t t) {
    uint32_t w = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) w |= ((t >> j) & 1u) << ((uint32_t)(d.posq >> (5 * j)) & 31u);
    return w;
}

#define TL_ROWS 64
#define TL_THREADS 1024
#define TL_WAVES (TL_THREADS / 64)

// all loads of a wave are issued before the first LDS store (2^NB / 16 granules in flight per wave)
template <int NB, int WAVES = TL_WAVES>
__device__ __forceinline__ void tile_load(uint32_t *lds, const uint32_t *__restrict__ Z, const tile_desc &d,
                                          uint32_t base_w, uint32_t rh, uint32_t wave, uint32_t lane) {
    constexpr int PER = (1 << NB) / WAVES;
    const uint32_t nslots = 1u << d.nb;
    const uint32_t *g = Z + (uint64_t)(base_w | tile_word(d, wave)) * ROWS + rh * TL_ROWS + lane;
    uint32_t v[PER];
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t s = wave + WAVES * i;
        v[i] = 0;
        if (s < nslots && (d.zero_bit >= 32 || !((s >> d.zero_bit) & 1u)))
            v[i] = g[(uint64_t)tile_word(d, WAVES * i) * ROWS];
    }
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t s = wave + WAVES * i;
        if (s < nslots) lds[s * TL_ROWS + lane] = v[i];
    }
}
// the same in two steps, for the persistent kernels that fetch the next tile while they work on the current one.
// The granules travel in one vector value (a loop-carried private ARRAY ends up in scratch memory).
template <int N>
struct tile_regs {
    typedef uint32_t type __attribute__((ext_vector_type(N)));
};
template <int NB>
__device__ __forceinline__ typename tile_regs<(1 << NB) / TL_WAVES>::type tile_fetch(
    const uint32_t *__restrict__ Z, const tile_desc &d, uint32_t base_w, uint32_t rh, uint32_t wave, uint32_t lane) {
    constexpr int PER = (1 << NB) / TL_WAVES;
    const uint32_t nslots = 1u << d.nb;
    // wave-uniform base (scalar registers) + one 32-bit per-lane byte offset shared by all PER loads: per-lane 64-bit
    // addresses for 32 granules in flight cost 64 VGPRs and spilled (k_top<.,9>: 13-16 VGPRs in scratch). Z < 4 GiB.
    const uint32_t voff = (((base_w | tile_word(d, wave)) * ROWS) + rh * TL_ROWS + lane) * 4u;
    // the granule offsets tile_word(d, 16 i) are loop invariants of the persistent kernels: hoisted, they (and those of
    // tile_store) sit in 128 SGPRs and spill into VGPR lanes. An opaque copy of the position word keeps them inside the
    // loop, rebuilt from five basis offsets (tile bits 4..8) with a few scalar ORs each.
    uint64_t pq = d.posq;
    asm volatile("" : "+s"(pq));
    uint32_t basis[5];
#pragma unroll
    for (int j = 0; j < 5; j++) basis[j] = 1u << ((uint32_t)(pq >> (5 * (4 + j))) & 31u);
    // which of the PER granules are not read (beyond the tile, or in the half that is known to be zero): ONE opaque mask,
    // tested bit by bit. As PER separate loop-invariant conditions they were hoisted out of the persistent loop into SGPR
    // pairs (k_top<.,9>: 320 SGPR spills), pushed on into VGPRs and from there to scratch -- and every scratch reload
    // waits for vmcnt(0), i.e. for the prefetched tile: k_top ran 1.8 ms per step instead of 1.6.
    uint32_t skip = PER * TL_WAVES > nslots ? ~0u << (nslots / TL_WAVES) : 0u;
    if (d.zero_bit < 4)
        skip |= ((wave >> d.zero_bit) & 1u) ? ~0u : 0u;
    else if (d.zero_bit < 32)
        skip |= d.zero_bit == 4 ? 0xAAAAAAAAu : d.zero_bit == 5 ? 0xCCCCCCCCu : d.zero_bit == 6 ? 0xF0F0F0F0u
                : d.zero_bit == 7 ? 0xFF00FF00u : 0xFFFF0000u;
    asm volatile("" : "+s"(skip));
    typename tile_regs<PER>::type v;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        uint32_t x = 0;
        if (!((skip >> i) & 1u)) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 5; j++)
                if ((i >> j) & 1) w |= basis[j];
            const char *ub = (const char *)Z + (uint64_t)w * (ROWS * 4);
            x = *(const uint32_t *)(ub + voff);
        }
        v[i] = x;
    }
    return v;
}
template <int NB>
__device__ __forceinline__ void tile_put(uint32_t *lds, typename tile_regs<(1 << NB) / TL_WAVES>::type v,
                                         const tile_desc &d, uint32_t wave, uint32_t lane) {
    constexpr int PER = (1 << NB) / TL_WAVES;
    uint32_t lim = (1u << d.nb) / TL_WAVES;  // slot wave + 16 i exists iff i < lim; opaque: see tile_fetch
    asm volatile("" : "+s"(lim));
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t s = wave + TL_WAVES * i;
        if ((uint32_t)i < lim) lds[s * TL_ROWS + lane] = v[i];
    }
}
template <int NB, int WAVES = TL_WAVES>
__device__ __forceinline__ void tile_store(const uint32_t *lds, uint32_t *__restrict__ Z, const tile_desc &d,
                                           uint32_t base_w, uint32_t rh, uint32_t wave, uint32_t lane) {
    constexpr int PER = (1 << NB) / WAVES;
    constexpr int LWV = WAVES == 16 ? 4 : WAVES == 8 ? 3 : WAVES == 4 ? 2 : WAVES == 2 ? 1 : 0;
    uint32_t lim = (1u << d.nb) >> LWV;  // slot wave + WAVES i exists iff i < lim; opaque: see tile_fetch
    asm volatile("" : "+s"(lim));
    const uint32_t voff = (((base_w | tile_word(d, wave)) * ROWS) + rh * TL_ROWS + lane) * 4u;
    uint64_t pq = d.posq;  // see tile_fetch
    asm volatile("" : "+s"(pq));
    uint32_t basis[9];
#pragma unroll
    for (int j = 0; j < 9 - LWV; j++) basis[j] = 1u << ((uint32_t)(pq >> (5 * (LWV + j))) & 31u);
#pragma unroll
    for (int i = 0; i < PER; i++) {
        const uint32_t s = wave + WAVES * i;
        if ((uint32_t)i < lim) {
            uint32_t w = 0;
#pragma unroll
            for (int j = 0; j < 9 - LWV; j++)
                if ((i >> j) & 1) w |= basis[j];
            *(uint32_t *)((char *)Z + (uint64_t)w * (ROWS * 4) + voff) = lds[s * TL_ROWS + lane];
        }
    }
}

// one work item of a butterfly level: outputs c0 .. c0+NO-1 (NO = min(K, 8)) of the butterfly whose x planes sit in
// slots xs .. xs+K-1 and y planes in ys .. ys+K-1.
//   forward: x ^= tau y; y ^= x      inverse: y ^= x; x ^= tau y      pruned (both ways): x ^= tau_lo y
// Results are returned in registers; the caller writes them after a barrier (other waves read the same slots).
template <int K, bool INV>
__device__ __forceinline__ void bfly_item(const uint32_t *lds, uint32_t xs, uint32_t ys, const uint32_t *__restrict__ TW,
                                          uint32_t q, uint32_t c0, bool pruned, uint32_t lane, uint32_t xn[8],
                                          uint32_t yn[8]) {
    constexpr int NO = K < 8 ? K : 8;
    uint32_t rows[NO];
    tw_rows<NO>(TW, q, c0, rows);
    uint32_t y[K];
#pragma unroll
    for (int j = 0; j < K; j++) y[j] = lds[(ys + j) * TL_ROWS + lane];
    if (INV && !pruned) {
#pragma unroll
        for (int j = 0; j < K; j++) y[j] ^= lds[(xs + j) * TL_ROWS + lane];
    }
    uint32_t acc[NO];
#pragma unroll
    for (int kk = 0; kk < NO; kk++) acc[kk] = lds[(xs + c0 + kk) * TL_ROWS + lane];
    // the NO x K strip in NO x NO blocks, each the matrix of a GF(2^NO) element: kara_acc
    kara_acc<NO, 0, 0>(acc, y, rows);
    if constexpr (K >= 16) kara_acc<NO, 0, 8>(acc, y + 8, rows);
    if constexpr (K >= 32) {
        kara_acc<NO, 0, 16>(acc, y + 16, rows);
        kara_acc<NO, 0, 24>(acc, y + 24, rows);
    }
#pragma unroll
    for (int kk = 0; kk < NO; kk++) xn[kk] = acc[kk];
    if (!pruned) {
        // y chunk: re-read (dynamic register indexing would spill)
#pragma unroll
        for (int kk = 0; kk < NO; kk++) {
            uint32_t yc = lds[(ys + c0 + kk) * TL_ROWS + lane];
            if (INV)
                yn[kk] = yc ^ lds[(xs + c0 + kk) * TL_ROWS + lane];
            else
                yn[kk] = yc ^ xn[kk];
        }
    }
}

template <bool INV, int WAVES = TL_WAVES>
__device__ __forceinline__ void tile_levels(uint32_t *lds, const tile_desc &d, uint32_t base_w,
                                            const uint32_t *__restrict__ TW, uint32_t wave, uint32_t lane) {
    for (uint32_t qi = 0; qi < d.nlev; qi++) {
        const uint32_t L = INV ? d.nlev - 1 - qi : qi;
        const uint32_t lv = d.lev[L];
        const uint32_t r = lv & 255u, tbit = (lv >> 8) & 255u, np = (lv >> 16) & 255u;
        const bool pruned = (lv >> 24) != 0;
        const uint32_t K = 1u << np;
        const uint32_t lnch = K > 8 ? np - 3 : 0;  // log2(output chunks per butterfly)
        const uint32_t nitems = (1u << (d.nb - 1 - np)) << lnch;
        const uint32_t NO = K < 8 ? K : 8;
        for (uint32_t it0 = 0; it0 < nitems; it0 += WAVES) {
            const uint32_t it = it0 + wave;
            const bool act = it < nitems;
            uint32_t xn[8], yn[8];
            uint32_t xs = 0, ys = 0, c0 = 0;
            if (act) {
                uint32_t bf = it >> lnch;
                c0 = (it & ((1u << lnch) - 1u)) << 3;
                uint32_t s = 0;
                for (uint32_t j = np; j < d.nb; j++)
                    if (j != tbit) {
                        s |= (bf & 1u) << j;
                        bf >>= 1;
                    }
                xs = s;
                ys = s | (1u << tbit);
                const uint32_t w = base_w | tile_word(d, s);
                const uint32_t q = ((1u << (d.m - 5)) | (w & ~d.pmw)) >> (d.m - r - 5);
                switch (np) {
                    case 2: bfly_item<4, INV>(lds, xs, ys, TW, q, c0, pruned, lane, xn, yn); break;
                    case 3: bfly_item<8, INV>(lds, xs, ys, TW, q, c0, pruned, lane, xn, yn); break;
                    case 4: bfly_item<16, INV>(lds, xs, ys, TW, q, c0, pruned, lane, xn, yn); break;
                    default: bfly_item<32, INV>(lds, xs, ys, TW, q, c0, pruned, lane, xn, yn); break;
                }
            }
            __syncthreads();
            if (act) {
#pragma unroll
                for (int kk = 0; kk < 8; kk++)
                    if (kk < (int)NO) {
                        lds[(xs + c0 + kk) * TL_ROWS + lane] = xn[kk];
                        if (!pruned) lds[(ys + c0 + kk) * TL_ROWS + lane] = yn[kk];
                    }
            }
            __syncthreads();
        }
    }
}

// whole-unit phases of the basis conversion (unit = 2^16 coefficients = 2048 words; pw and half are multiples of
// a unit): plain XORs between granules with the same offset in different units. They commute with the phases
// that are local to a unit, so they run in the same LDS tile as the top butterfly levels (whose word bits are
// unit-index bits). The host flattens them into a list of (dst slot, src slot) moves grouped in hazard-free steps
// (fold / shift of each phase); a 16-lane group moves one granule (64 rows x 4 B) per list entry.
struct unit_phases {
    uint32_t nsteps;
    uint32_t step_end[UP_MAX_STEPS];  // list index one past the last move of step s
    const uint32_t *list;             // device memory: PH_ENTRY words
};

// A list entry is one index j of a phase with BOTH of its operations, run by the same 16-lane group (see PH_ENTRY):
// forward  h = L[a]; if (fold) { h ^= L[b]; L[a] = h; }  L[c] ^= h      (fold, then shift)
// inverse  h = L[a]; L[c] ^= h; if (fold) L[a] = h ^ L[b]              (shift, then fold)
// a = H[j], b = H[j + D] (read only in this phase), c = lo[pw + j]: no other entry of the phase touches a or c, so a
// phase is ONE hazard-free step (one barrier), not two.
#define PH_ENTRY(a, c, b, fold) ((a) | ((c) << 10) | ((b) << 20) | ((fold) ? 1u << 30 : 0u))
template <bool INV>
__device__ __forceinline__ void tile_unit_phases(uint32_t *lds, const unit_phases &up, const uint32_t *list) {
    u4 *l4 = (u4 *)lds;
    const uint32_t rq = threadIdx.x & 15, g = threadIdx.x >> 4;
    uint32_t start = 0;
    for (uint32_t st = 0; st < up.nsteps; st++) {
        const uint32_t end = up.step_end[st];
        for (uint32_t it = start + g; it < end; it += TL_THREADS / 16) {
            const uint32_t e = list[it];
            const uint32_t a = (e & 1023u) * 16 + rq, c = ((e >> 10) & 1023u) * 16 + rq, b = ((e >> 20) & 1023u) * 16 + rq;
            u4 h = l4[a];
            if (e >> 30) {
                const u4 gg = l4[b];
                if (!INV) h ^= gg;
                l4[a] = INV ? h ^ gg : h;
            }
            l4[c] ^= h;
        }
        start = end;
        __syncthreads();
    }
}

// word-level butterfly passes (top / mid levels): gather, [unit phases,] levels, scatter
template <bool INV, int NB>
__global__ void __launch_bounds__(TL_THREADS)
    k_tile_fft(uint32_t *__restrict__ Z, tile_desc d, const uint32_t *__restrict__ TW, unit_phases up) {
    __shared__ uint32_t lds[(1 << NB) * TL_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t rh = tile_rh(d, blockIdx.x);
    const uint32_t base_w = tile_base(d, tile_no(d, blockIdx.x));
    tile_load<NB>(lds, Z, d, base_w, rh, wave, lane);
    __syncthreads();
    if (!INV && up.nsteps) tile_unit_phases<false>(lds, up, up.list);
    tile_levels<INV>(lds, d, base_w, TW, wave, lane);
    if (INV && up.nsteps) tile_unit_phases<true>(lds, up, up.list);
    tile_store<NB>(lds, Z, d, base_w, rh, wave, lane);
}

// The extra 32-plane word-level passes of m >= 26 (one or two levels over the whole array): a 128-slot tile for BOTH
// row halves in a workgroup of 8 waves. A wave owns one work item (butterfly x 8 output planes) of a level and applies
// it to rows 0..63 and rows 64..127 with the SAME scalar masks: these levels are bound by the scalar unit (one
// s_bfe_i32 per XAND), so sharing the masks between the row halves halves their cost; the workgroup also touches whole
// 512-B granules. 64 KiB of LDS and 8 waves: two workgroups per CU overlap each other's loads and stores.
#define W32_THREADS 512
template <bool INV>
__device__ __forceinline__ void w32_item(const uint32_t *l0, const uint32_t *l1, uint32_t xs, uint32_t ys,
                                         const uint32_t *__restrict__ TW, uint32_t q, uint32_t c0, uint32_t lane,
                                         uint32_t xn0[8], uint32_t yn0[8], uint32_t xn1[8], uint32_t yn1[8]) {
    uint32_t rows[8];
    tw_rows<8>(TW, q, c0, rows);
    uint32_t y0[32], y1[32];
#pragma unroll
    for (int j = 0; j < 32; j++) {
        y0[j] = l0[(ys + j) * TL_ROWS + lane];
        y1[j] = l1[(ys + j) * TL_ROWS + lane];
    }
    if (INV) {
#pragma unroll
        for (int j = 0; j < 32; j++) {
            y0[j] ^= l0[(xs + j) * TL_ROWS + lane];
            y1[j] ^= l1[(xs + j) * TL_ROWS + lane];
        }
    }
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        xn0[kk] = l0[(xs + c0 + kk) * TL_ROWS + lane];
        xn1[kk] = l1[(xs + c0 + kk) * TL_ROWS + lane];
    }
    // the 8 x 32 strip is four 8 x 8 blocks, each the matrix of a GF(2^8) element: kara_acc (the scalar masks of the two
    // row halves are the same expressions and are formed once)
    kara_acc<8, 0, 0>(xn0, y0, rows);
    kara_acc<8, 0, 0>(xn1, y1, rows);
    kara_acc<8, 0, 8>(xn0, y0 + 8, rows);
    kara_acc<8, 0, 8>(xn1, y1 + 8, rows);
    kara_acc<8, 0, 16>(xn0, y0 + 16, rows);
    kara_acc<8, 0, 16>(xn1, y1 + 16, rows);
    kara_acc<8, 0, 24>(xn0, y0 + 24, rows);
    kara_acc<8, 0, 24>(xn1, y1 + 24, rows);
    // y chunk: re-read (dynamic register indexing would spill)
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
        const uint32_t a0 = l0[(ys + c0 + kk) * TL_ROWS + lane], a1 = l1[(ys + c0 + kk) * TL_ROWS + lane];
        if (INV) {
            yn0[kk] = a0 ^ l0[(xs + c0 + kk) * TL_ROWS + lane];
            yn1[kk] = a1 ^ l1[(xs + c0 + kk) * TL_ROWS + lane];
        } else {
            yn0[kk] = a0 ^ xn0[kk];
            yn1[kk] = a1 ^ xn1[kk];
        }
    }
}

template <bool INV>
__global__ void __launch_bounds__(W32_THREADS)
    k_tile_fft_pair(uint32_t *__restrict__ Z, tile_desc d, const uint32_t *__restrict__ TW) {
    __shared__ uint32_t lds2[2][(1 << 7) * TL_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *l0 = lds2[0], *l1 = lds2[1];
    const uint32_t base_w = tile_base(d, blockIdx.x);
    tile_load<7, 8>(l0, Z, d, base_w, 0, wave, lane);
    tile_load<7, 8>(l1, Z, d, base_w, 1, wave, lane);
    __syncthreads();
    // d.nb == 7 and every level has 32 planes (np = 5, none pruned): 2 butterflies x 4 output chunks = 8 items
    for (uint32_t qi = 0; qi < d.nlev; qi++) {
        const uint32_t lv = d.lev[INV ? d.nlev - 1 - qi : qi];
        const uint32_t r = lv & 255u, tbit = (lv >> 8) & 255u;
        uint32_t bf = wave >> 2;
        const uint32_t c0 = (wave & 3u) << 3;
        uint32_t sl = 0;
        for (uint32_t j = 5; j < 7; j++)
            if (j != tbit) {
                sl |= (bf & 1u) << j;
                bf >>= 1;
            }
        const uint32_t xs = sl, ys = sl | (1u << tbit);
        const uint32_t w = base_w | tile_word(d, sl);
        const uint32_t q = ((1u << (d.m - 5)) | (w & ~d.pmw)) >> (d.m - r - 5);
        uint32_t xn0[8], yn0[8], xn1[8], yn1[8];
        w32_item<INV>(l0, l1, xs, ys, TW, q, c0, lane, xn0, yn0, xn1, yn1);
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
            l0[(xs + c0 + kk) * TL_ROWS + lane] = xn0[kk];
            l0[(ys + c0 + kk) * TL_ROWS + lane] = yn0[kk];
            l1[(xs + c0 + kk) * TL_ROWS + lane] = xn1[kk];
            l1[(ys + c0 + kk) * TL_ROWS + lane] = yn1[kk];
        }
        __syncthreads();
    }
    tile_store<7, 8>(l0, Z, d, base_w, 0, wave, lane);
    tile_store<7, 8>(l1, Z, d, base_w, 1, wave, lane);
}

// ------------------------------------------------------------------------------------------------
// the five intra-word levels (i = 4..0 forward, 0..4 inverse) on one 32-plane value word W[32]. Level i pairs bit
// positions p and p + 2^i; the twiddle of position p is point(qw >> i) ^ point(p >> i) with
// qw = 2^m | (position of the word's bit 0, pruned bits cleared): a wave-uniform part (always a full GF(2^32)
// element here) and a per-position part in GF(2^8) (compile-time masks QC_MPOS).
__device__ __forceinline__ uint32_t lowmask(int i) {
    return i == 0 ? 0x55555555u : i == 1 ? 0x33333333u : i == 2 ? 0x0F0F0F0Fu : i == 3 ? 0x00FF00FFu : 0x0000FFFFu;
}

// a ? (c-selected) : bitfield insert  (c & a) | (~c & b)
#define BFI(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0xE4)

// One level on the value word W[32]. Only half of the bit positions of a plane hold y operands, so planes j and
// j + 16 are packed into one word (y_j at the x positions, y_(j+16) at the y positions) and the twiddle matrix is
// applied with mask words precomputed per (value word, level) by k_build_lowmasks: bit p of a mask = one entry of the
// matrix of tau(p) at x positions, of tau(p - 2^I) at y positions, uniform + per-position twiddle parts already combined.
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));

// One intra-word level with its masks read from LDS as VGPR operands (ds_read_b128, same address in every lane): no
// scalar-memory stream, whose two rows in flight cannot cover the latency of a table that does not fit any cache.
// The LDS pipe delivers one such read per ~5 cycles and CU, i.e. the level is bound by the NUMBER of mask words, so the
// 32 x 32 twiddle matrix of a position is applied Karatsuba-style over the tower GF(2^32) = GF(2^16)[t], t^2 = t + zeta:
// in blocks of 16 planes it is [[A, X], [B, A + B]], hence with y = (y_lo, y_hi), s = y_lo + y_hi
//     out_lo = A y_lo + X y_hi                    (set 1: 16 x 16 mask words, y_lo at the x positions, y_hi at the y positions)
//     out_hi = (A + B) s + A y_lo                 (set 2:  8 x 16 mask words, s_j at the x positions, s_(j+8) at the y positions)
// 384 mask words per (value word, level) instead of 512 (k_build_lowmasks), + ~90 plain VALU ops that the LDS bound hides.
// Plain layout of a block: 96 uint4 = 384 words, set 1 = M[4 j + kq], j < 16; set 2 = M[64 + 4 j' + kq], j' < 8 (levels);
// A | X rows 0..7 and X rows 8..15 | A + B (the two blocks of Amat). The recursion below never reads the b blocks, so
// the tables in memory hold only the uint4 that are read, in plain order (d_LOWCMAP[type][plain uint4] = packed index or
// -1; type 0 = level block: 60 of 96, 1 / 2 = first / second Amat block: 48 and 60): 37 % less mask traffic, one
// 16-byte load per lane and block.
#define LOW_BLK_PLAIN 384
#define LOW_BLK_WORDS 240  // packed words of a level block
#define AMAT_PLAIN 768
#define AMAT_B0_WORDS 192  // packed words of the first Amat block
#define AMAT_WORDS 432     // packed words per value word
#define LOW_SLOT 60        // uint4 per ring slot
static __device__ const int8_t d_LOWCMAP[3][96] = {
    {0, -1, -1, -1, 1, -1, -1, -1, 2, -1, -1, -1, 3, -1, -1, -1, 4, 5, -1, -1, 6, 7, -1, -1, 8, 9, -1, -1, 10, 11, -1, -1, 12, -1, 13, -1, 14, -1, 15, -1, 16, -1, 17, -1, 18, -1, 19, -1, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, -1, 37, -1, 38, -1, 39, -1, 40, -1, 41, -1, 42, -1, 43, -1, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59},
    {0, -1, -1, -1, 1, -1, -1, -1, 2, -1, -1, -1, 3, -1, -1, -1, 4, 5, -1, -1, 6, 7, -1, -1, 8, 9, -1, -1, 10, 11, -1, -1, 12, -1, 13, -1, 14, -1, 15, -1, 16, -1, 17, -1, 18, -1, 19, -1, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, -1, -1, -1, 37, -1, -1, -1, 38, -1, -1, -1, 39, -1, -1, -1, 40, 41, -1, -1, 42, 43, -1, -1, 44, 45, -1, -1, 46, 47, -1, -1},
    {0, -1, 1, -1, 2, -1, 3, -1, 4, -1, 5, -1, 6, -1, 7, -1, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, -1, -1, -1, 25, -1, -1, -1, 26, -1, -1, -1, 27, -1, -1, -1, 28, 29, -1, -1, 30, 31, -1, -1, 32, 33, -1, -1, 34, 35, -1, -1, 36, -1, 37, -1, 38, -1, 39, -1, 40, -1, 41, -1, 42, -1, 43, -1, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59}};
struct IdMap {
    static __device__ __forceinline__ int at(int i) { return i; }
};
template <int TYPE, int OFF>
struct LowMap {
    static __device__ __forceinline__ int at(int i) { return d_LOWCMAP[TYPE][OFF + i]; }
};
// acc ^= B v for the N x N block B = rows K0.., columns J0.. of a mask set M[4 j + k / 4] (16 output planes per j) whose
// matrix (both packed halves, if packed) is that of GF(2^N) elements: Karatsuba down to 4 x 4 on the SAME table,
// [[a, x], [b, a + b]] in blocks; the b blocks are never read: 36 broadcast reads per 16 x 16 block instead of 64
// (12 instead of 16 per 8 x 8). acc and v are relative to the block.
#ifndef KARA_LDS_BASE
#define KARA_LDS_BASE 4
#endif
template <int N, int K0, int J0, int BASE = KARA_LDS_BASE, class MAP = IdMap>
__device__ __forceinline__ void kara_lds(uint32_t *acc, const uint32_t *v, const uint4 *M) {
    if constexpr (N <= BASE) {
#pragma unroll
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int kq = 0; kq < N / 4; kq++) {
                const uint4 m = M[MAP::at((J0 + j) * 4 + K0 / 4 + kq)];
                acc[4 * kq + 0] = XAND(acc[4 * kq + 0], v[j], m.x);
                acc[4 * kq + 1] = XAND(acc[4 * kq + 1], v[j], m.y);
                acc[4 * kq + 2] = XAND(acc[4 * kq + 2], v[j], m.z);
                acc[4 * kq + 3] = XAND(acc[4 * kq + 3], v[j], m.w);
            }
        }
    } else {
        constexpr int H = N / 2;
        uint32_t t1[H], sm[H];
#pragma unroll
        for (int k = 0; k < H; k++) t1[k] = 0, sm[k] = v[k] ^ v[k + H];
        kara_lds<H, K0, J0, BASE, MAP>(t1, v, M);
        kara_lds<H, K0, J0 + H, BASE, MAP>(acc, v + H, M);
        kara_lds<H, K0 + H, J0 + H, BASE, MAP>(acc + H, sm, M);
#pragma unroll
        for (int k = 0; k < H; k++) acc[k] ^= t1[k], acc[k + H] ^= t1[k];
    }
}
template <int I, bool INV>
__device__ __forceinline__ void low_level_lds(uint32_t W[32], const uint4 *M) {
    const uint32_t A = lowmask(I);
    constexpr int SH = 1 << I;
    if (INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] ^= (W[k] & A) << SH;
    }
    uint32_t P[16];  // y_j at the x positions, y_(j+16) at the y positions
#pragma unroll
    for (int j = 0; j < 16; j++) P[j] = BFI(W[j] >> SH, W[j + 16], A);
    uint32_t a1[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a1[k] = 0;
    kara_lds<16, 0, 0, KARA_LDS_BASE, LowMap<0, 0>>(a1, P, M);  // x positions: A y_lo, y positions: X y_hi
    uint32_t S[8];
#pragma unroll
    for (int j = 0; j < 8; j++) S[j] = BFI(P[j] ^ (P[j] >> SH), P[j + 8] ^ (P[j + 8] << SH), A);
    uint32_t a3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a3[k] = 0;
    // x + y positions: (A + B) s; per 8 output planes the two packed 8 x 8 halves are matrices of GF(2^8) elements
    kara_lds<8, 0, 0, KARA_LDS_BASE, LowMap<0, 64>>(a3, S, M);
    kara_lds<8, 8, 0, KARA_LDS_BASE, LowMap<0, 64>>(a3 + 8, S, M);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        W[k] = XAND(W[k], a1[k] ^ (a1[k] >> SH), A);
        W[k + 16] = XAND(W[k + 16], a3[k] ^ (a3[k] >> SH) ^ a1[k], A);
    }
    if (!INV) {
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] ^= (W[k] & A) << SH;
    }
}
// pointwise multiply by FFT(a), same decomposition, the 768 mask words of a value word as two blocks of 384:
// block 0 = A | X[0..7], block 1 = X[8..15] | (A + B)
__device__ __forceinline__ void amat_block0(const uint32_t W[32], const uint4 *M, uint32_t t1[16], uint32_t tt[8]) {
#pragma unroll
    for (int k = 0; k < 16; k++) t1[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) tt[k] = 0;
    kara_lds<16, 0, 0, KARA_LDS_BASE, LowMap<1, 0>>(t1, W, M);         // A w_lo
    kara_lds<8, 0, 0, KARA_LDS_BASE, LowMap<1, 64>>(tt, W + 16, M);    // the a block of X times the low half of w_hi
}
__device__ __forceinline__ void amat_block1(uint32_t W[32], const uint4 *M, const uint32_t t1[16], const uint32_t tt[8]) {
    uint32_t t2[16], sx[8];
#pragma unroll
    for (int k = 0; k < 8; k++) t2[k] = tt[k], t2[k + 8] = tt[k], sx[k] = W[16 + k] ^ W[24 + k];
    kara_lds<8, 0, 0, KARA_LDS_BASE, LowMap<2, 0>>(t2, W + 24, M);   // x block of X
    kara_lds<8, 8, 0, KARA_LDS_BASE, LowMap<2, 0>>(t2 + 8, sx, M);   // (a + b) block of X
    uint32_t sm[16], t3[16];
#pragma unroll
    for (int j = 0; j < 16; j++) sm[j] = W[j] ^ W[j + 16], t3[j] = 0;
    kara_lds<16, 0, 0, KARA_LDS_BASE, LowMap<2, 32>>(t3, sm, M);  // (A + B)(w_lo + w_hi)
#pragma unroll
    for (int k = 0; k < 16; k++) W[k] = t1[k] ^ t2[k], W[k + 16] = t3[k] ^ t1[k];
}
// mask blocks of a value word in the order the fused kernel consumes them (2 KiB each): forward levels 4..0, the two
// halves of Amat, inverse levels 0..4. Word offset into LM (levels) or Amat.
__device__ __forceinline__ constexpr int lowblk_level(int b) { return b < 5 ? 4 - b : b - 7; }
__device__ __forceinline__ constexpr bool lowblk_is_amat(int b) { return b == 5 || b == 6; }

// ------------------------------------------------------------------------------------------------
// K4t: the top pass: whole-unit conversion phases + levels 1..8 (1 and 2 only relabel bits as planes).
// Tile bits: t0..t3 = planes (levels 1, 2, 4, 8), t4..t7 = levels 3, 5, 6, 7, t8 = an extra unit-index bit (m = 25).
// The six real levels run as three register rounds of two levels each; in a round every wave owns whole groups of
// slots (all planes x the two pair bits of the round), so one barrier per round is enough.
template <int K, bool INV>
__device__ __forceinline__ void reg_bfly(uint32_t x[K], uint32_t y[K], const uint32_t *__restrict__ TW, uint32_t q,
                                         bool pruned) {
    uint32_t rows[K];
    tw_rows<K, 2>(TW, q, 0, rows);  // levels 3 .. 8: q < 2^9
    if (INV && !pruned) {
#pragma unroll
        for (int k = 0; k < K; k++) y[k] ^= x[k];
    }
    kara_acc<K, 0, 0>(x, y, rows);
    if (!INV && !pruned) {
#pragma unroll
        for (int k = 0; k < K; k++) y[k] ^= x[k];
    }
}

// one round: group slots = base | p | (a << ba) | (b << bb), p < K planes; level 1 pairs bit a, level 2 pairs bit b
// (forward order). q1(b) and q2(a) give the twiddle indices (they may depend on the other pair bit of the round).
template <int K, bool INV, class Q1, class Q2>
__device__ __forceinline__ void top_round(uint32_t *lds, uint32_t base, uint32_t ba, uint32_t bb, bool pruned2,
                                          const uint32_t *__restrict__ TW, uint32_t lane, Q1 q1, Q2 q2) {
    uint32_t v[2][2][K];  // [b][a][plane]
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int p = 0; p < K; p++) v[b][a][p] = lds[(base | p | (a << ba) | (b << bb)) * TL_ROWS + lane];
    if (!INV) {
#pragma unroll
        for (int b = 0; b < 2; b++) reg_bfly<K, false>(v[b][0], v[b][1], TW, q1(b), false);
#pragma unroll
        for (int a = 0; a < 2; a++) reg_bfly<K, false>(v[0][a], v[1][a], TW, q2(a), pruned2);
    } else {
#pragma unroll
        for (int a = 0; a < 2; a++) reg_bfly<K, true>(v[0][a], v[1][a], TW, q2(a), pruned2);
#pragma unroll
        for (int b = 0; b < 2; b++) reg_bfly<K, true>(v[b][0], v[b][1], TW, q1(b), false);
    }
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int p = 0; p < K; p++) lds[(base | p | (a << ba) | (b << bb)) * TL_ROWS + lane] = v[b][a][p];
}

template <bool INV>
__device__ __forceinline__ void top_levels(uint32_t *lds, uint32_t nb, const uint32_t *__restrict__ TW, uint32_t wave,
                                           uint32_t lane) {
    const uint32_t nx = nb - 8;  // extra tile bits above t7
    for (int rnd = 0; rnd < 3; rnd++) {
        const int R = INV ? 2 - rnd : rnd;
        if (R == 0) {
            // levels 3 (pair bit t4) and 4 (pair bit t2, pruned), K = 4; free bits: t3, t5, t6, t7, extras
            const uint32_t ngroups = 16u << nx;
            for (uint32_t g = wave; g < ngroups; g += TL_WAVES) {
                const uint32_t base = ((g & 1u) << 3) | ((g >> 1) << 5);
                top_round<4, INV>(lds, base, 4, 2, true, TW, lane, [&](int) { return 8u; },
                                  [&](int a) { return 16u | ((uint32_t)a << 1); });
            }
        } else if (R == 1) {
            // levels 5 (t5) and 6 (t6), K = 8; free bits: t3, t4, t7, extras
            const uint32_t ngroups = 8u << nx;
            for (uint32_t g = wave; g < ngroups; g += TL_WAVES) {
                const uint32_t j3 = (g >> 1) & 1u;
                const uint32_t base = ((g & 1u) << 3) | (j3 << 4) | ((g >> 2) << 7);
                top_round<8, INV>(lds, base, 5, 6, false, TW, lane, [&](int) { return 32u | (j3 << 2); },
                                  [&](int a) { return 64u | (j3 << 3) | ((uint32_t)a << 1); });
            }
        } else {
            // levels 7 (t7) and 8 (t3, pruned), K = 8; free bits: t4, t5, t6, extras
            const uint32_t ngroups = 8u << nx;
            for (uint32_t g = wave; g < ngroups; g += TL_WAVES) {
                const uint32_t j3 = g & 1u, j5 = (g >> 1) & 1u, j6 = (g >> 2) & 1u;
                const uint32_t base = ((g & 7u) << 4) | ((g >> 3) << 8);
                const uint32_t qc = (j3 << 4) | (j5 << 2) | (j6 << 1);
                top_round<8, INV>(lds, base, 7, 3, true, TW, lane, [&](int) { return 128u | qc; },
                                  [&](int a) { return 256u | (qc << 1) | ((uint32_t)a << 1); });
            }
        }
        __syncthreads();
    }
}

// Persistent (one workgroup per CU, tiles idx = blockIdx.x, + gridDim.x, ...): the tile fills the CU's LDS, so the
// next tile's granules are fetched into registers while the current tile is processed; the move list of the unit
// phases is staged in LDS once (a global load in the middle of a tile would drain the prefetch).
#define TOP_MAX_MOVES 3584
template <bool INV, int NB>
__global__ void __launch_bounds__(TL_THREADS)
    k_top(uint32_t *__restrict__ Z, tile_desc d, const uint32_t *__restrict__ TW, unit_phases up, uint32_t ntiles) {
    __shared__ uint32_t lds[(1 << NB) * TL_ROWS];
    __shared__ uint32_t mvl[TOP_MAX_MOVES];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t idx = blockIdx.x;
    if (idx >= ntiles) return;
    if (up.nsteps)
        for (uint32_t i = threadIdx.x; i < up.step_end[up.nsteps - 1]; i += TL_THREADS) mvl[i] = up.list[i];
    typename tile_regs<(1 << NB) / TL_WAVES>::type v =
        tile_fetch<NB>(Z, d, tile_base(d, tile_no(d, idx)), tile_rh(d, idx), wave, lane);
#pragma unroll 1
    for (; idx < ntiles; idx += gridDim.x) {
        const uint32_t rh = tile_rh(d, idx);
        const uint32_t base_w = tile_base(d, tile_no(d, idx));
        tile_put<NB>(lds, v, d, wave, lane);
        __syncthreads();
        {
            const uint32_t nidx = idx + gridDim.x < ntiles ? idx + gridDim.x : idx;
            v = tile_fetch<NB>(Z, d, tile_base(d, tile_no(d, nidx)), tile_rh(d, nidx), wave, lane);
        }
        if (!INV && up.nsteps && !(QC_EXP & 128)) tile_unit_phases<false>(lds, up, mvl);
        if (!(QC_EXP & 256)) top_levels<INV>(lds, d.nb, TW, wave, lane);
        if (INV && up.nsteps && !(QC_EXP & 128)) tile_unit_phases<true>(lds, up, mvl);
        tile_store<NB>(lds, Z, d, base_w, rh, wave, lane);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K4m: the 16-plane passes (levels 9..12 and 13..16). Tile = 16 values x 16 planes x 64 rows, slot = lv * 16 + plane,
// lv bit b <-> level rmin + b. One wave per butterfly and level (8 waves), all 16 planes in registers, twiddle
// matrices as precomputed mask words MT[outer][b][butterfly][j][k] (k_build_midmasks) staged in LDS (see k_mid).
#define MD_THREADS 512
#define MD_WAVES (MD_THREADS / 64)

__device__ __forceinline__ uint32_t pext32(uint32_t v, uint32_t mask) {
    uint32_t r = 0, o = 0;
    for (uint32_t b = 0; b < 27; b++)
        if ((mask >> b) & 1u) r |= ((v >> b) & 1u) << o++;
    return r;
}
__device__ __forceinline__ uint32_t pdep32(uint32_t v, uint32_t mask) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < 27; b++)
        if ((mask >> b) & 1u) {
            r |= (v & 1u) << b;
            v >>= 1;
        }
    return r;
}

// Persistent, two workgroups per CU (tiles idx = blockIdx.x, + gridDim.x, ...): the next tile's 32 granules per
// thread are fetched while the four levels of the current one run.
template <bool INV, bool PRUNE_LAST>
__device__ __forceinline__ void mid_body(uint32_t *__restrict__ Z, const tile_desc &d, uint32_t mask_hi,
                                         const uint32_t *__restrict__ MT, uint32_t ntiles, uint32_t *__restrict__ ctr) {
    __shared__ uint32_t lds[256 * TL_ROWS];
    __shared__ uint32_t s_next;
    constexpr int PER = 256 / MD_WAVES;
    typedef typename tile_regs<PER>::type regs_t;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto fetch = [&](uint32_t idx) {
        const uint32_t *g = Z + (uint64_t)(tile_base(d, tile_no(d, idx)) | tile_word(d, wave)) * ROWS + tile_rh(d, idx) * TL_ROWS + lane;
        regs_t v;
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = g[(uint64_t)tile_word(d, MD_WAVES * i) * ROWS];
        return v;
    };
    // a workgroup takes BOTH row halves of a tile back to back: the second half finds the tile's masks in L2
    // (d.rhalves == 1: one tile index per gather tile).
    // Which tile pair comes next is drawn from a counter (*ctr, zero at launch), not fixed by blockIdx: of the two
    // workgroups that share a CU the hardware serves the older one's waves first, so with equal shares one finishes its
    // list early and the other runs the rest alone, with half the waves the CU needs to overlap its phases
    // (tools/micro/cu_spread.hip). The first two pairs of a workgroup are fixed (blockIdx.x, + gridDim.x), each further
    // one is drawn one pair ahead, so the draw's latency never sits in front of a prefetch.
    const uint32_t RH = d.rhalves;
    const uint32_t npairs = ntiles / RH;
    if (blockIdx.x >= npairs) return;
    uint32_t cur = blockIdx.x, nxt = blockIdx.x + gridDim.x;
    const uint32_t bf = wave;
    regs_t v = fetch(RH * cur);
#pragma unroll 1
    while (cur < npairs) {
        uint32_t drawn = 0;
        if (threadIdx.x == 0) drawn = 2 * gridDim.x + atomicAdd(ctr, 1u);
#pragma unroll 1
        for (uint32_t h = 0; h < RH; h++) {
            const uint32_t idx = RH * cur + h;
            const uint32_t rh = tile_rh(d, idx);
            const uint32_t base_w = tile_base(d, tile_no(d, idx));
            const uint32_t outer = pext32(base_w, mask_hi);
#pragma unroll
            for (int i = 0; i < PER; i++) lds[(wave + MD_WAVES * i) * TL_ROWS + lane] = v[i];
            __syncthreads();
            // the 1 KiB of masks of each of the four levels, one 16-byte piece per lane, fetched now and parked in registers;
            // during a level they are staged in the wave's own x slots (free while x and y are in registers) and read back as
            // VGPR operands: no scalar-memory stream whose latency the level would expose
            // (one vector value, components picked by constant index: as an array of uint4 it lived in scratch memory, 64 B
            // written and read back per thread and tile = +50 % HBM writes on this kernel, profiles/r1_v10_pmc.csv)
            u32x16 mk;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const uint4 t = *(const uint4 *)(MT + ((((uint64_t)outer * 4 + b) * 8 + bf) << 8) + lane * 4);
                mk[4 * b + 0] = t.x, mk[4 * b + 1] = t.y, mk[4 * b + 2] = t.z, mk[4 * b + 3] = t.w;
            }
            // the prefetch of the next tile comes AFTER the mask loads: vector loads return in order, so waiting for masks
            // issued behind it would wait for the whole prefetch (vmcnt(0)) before the first level
            {
                const uint32_t nidx = h + 1 < RH ? idx + 1 : nxt < npairs ? RH * nxt : idx;
                v = fetch(nidx);
            }
#pragma unroll
            for (int q = 0; q < ((QC_EXP & 16) ? 0 : 4); q++) {
                const int b = INV ? 3 - q : q;
                const bool pruned = PRUNE_LAST && b == 3;
                const uint32_t lvx = ((bf >> b) << (b + 1)) | (bf & ((1u << b) - 1u));
                uint32_t *px = lds + (lvx << 4) * TL_ROWS + lane;
                uint32_t *py = px + (16u << b) * TL_ROWS;
                uint32_t x[16], y[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    x[k] = px[k * TL_ROWS];
                    y[k] = py[k * TL_ROWS];
                }
                if (INV && !pruned) {
#pragma unroll
                    for (int k = 0; k < 16; k++) y[k] ^= x[k];
                }
                {
                    // x ^= tau (x) y, Karatsuba-style (kara_lds) with 8 x 8 blocks (48 broadcast reads instead of 64) in the forward
                    // kernel and 4 x 4 blocks (36) in the inverse one: forward with 4 x 4 blocks needs 129 VGPRs, one too many
                    // for two workgroups per CU (1.50 ms per step instead of 1.17; forcing 128 by attribute: 1.22)
                    uint4 *R = (uint4 *)(lds + (lvx << 4) * TL_ROWS);  // this butterfly's x slots: 4 KiB, owned by this wave
                    R[lane] = make_uint4(mk[4 * b + 0], mk[4 * b + 1], mk[4 * b + 2], mk[4 * b + 3]);
                    kara_lds<16, 0, 0, INV ? 4 : 8>(x, y, R);
                }
#pragma unroll
                for (int k = 0; k < 16; k++) px[k * TL_ROWS] = x[k];
                if (!pruned) {
#pragma unroll
                    for (int k = 0; k < 16; k++) py[k * TL_ROWS] = INV ? y[k] : y[k] ^ x[k];
                }
                if (!(QC_EXP & 64)) __syncthreads();
            }
            {
                uint32_t *g = Z + (uint64_t)(base_w | tile_word(d, wave)) * ROWS + rh * TL_ROWS + lane;
#pragma unroll
                for (int i = 0; i < PER; i++)
                    g[(uint64_t)tile_word(d, MD_WAVES * i) * ROWS] = lds[(wave + MD_WAVES * i) * TL_ROWS + lane];
            }
            if (h == 0 && threadIdx.x == 0) s_next = drawn;
            __syncthreads();
        }
        cur = nxt;
        nxt = __builtin_amdgcn_readfirstlane(s_next);
    }
}
// the inverse kernel (4 x 4 Karatsuba blocks) lands at 131-135 VGPRs by itself: held at 128 so that two workgroups share a CU
// (the same attribute costs the forward kernel 13 %)
template <bool INV, bool PRUNE_LAST>
__global__ void __launch_bounds__(MD_THREADS)
    k_mid(uint32_t *__restrict__ Z, tile_desc d, uint32_t mask_hi, const uint32_t *__restrict__ MT, uint32_t ntiles,
          uint32_t *__restrict__ ctr) {
    mid_body<INV, PRUNE_LAST>(Z, d, mask_hi, MT, ntiles, ctr);
}
template <bool PRUNE_LAST>
__global__ void __launch_bounds__(MD_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
    k_mid_inv(uint32_t *__restrict__ Z, tile_desc d, uint32_t mask_hi, const uint32_t *__restrict__ MT, uint32_t ntiles,
              uint32_t *__restrict__ ctr) {
    mid_body<true, PRUNE_LAST>(Z, d, mask_hi, MT, ntiles, ctr);
}

// MT[((outer * 4 + b) * 8 + bf)][j][k] = all-ones iff output plane k of (twiddle (x) y) takes input plane j
__global__ void k_build_midmasks(tile_desc d, uint32_t mask_hi, uint32_t rmin, uint32_t nouter,
                                 const uint32_t *__restrict__ TW, uint32_t *__restrict__ MT) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;  // (outer, b, bf, j)
    if (idx >= nouter * 4 * 8 * 16) return;
    const uint32_t j = idx & 15, bf = (idx >> 4) & 7, b = (idx >> 7) & 3, outer = idx >> 9;
    const uint32_t lvx = ((bf >> b) << (b + 1)) | (bf & ((1u << b) - 1u));
    const uint32_t w = pdep32(outer, mask_hi) | tile_word(d, lvx << 4);
    const uint32_t r = rmin + b;
    const uint32_t q = ((1u << (d.m - 5)) | (w & ~d.pmw)) >> (d.m - r - 5);
    uint32_t rows[16];
    tw_rows<16>(TW, q, 0, rows);
    for (uint32_t k = 0; k < 16; k++) MT[(uint64_t)idx * 16 + k] = 0u - ((rows[k] >> j) & 1u);
}

// the 32-plane word levels of the low tile (slot = lv * 32 + plane, lv bit b <-> level m - 4 - nl + b): one wave per
// (butterfly, half of the output planes); results stay in registers across the barrier that separates the reads
// of a level from its writes.
template <bool INV, int C0>
__device__ __forceinline__ void low_word_item(const uint32_t *lds, uint32_t xs, uint32_t ys, const uint32_t *__restrict__ TW,
                                              uint32_t q, uint32_t lane, uint32_t xn[16], uint32_t yn[16]) {
    uint32_t rows[16];
    tw_rows<16>(TW, q, C0, rows);
    uint32_t y[32];
#pragma unroll
    for (int j = 0; j < 32; j++) y[j] = lds[(ys + j) * TL_ROWS + lane];
    if (INV) {
#pragma unroll
        for (int j = 0; j < 32; j++) y[j] ^= lds[(xs + j) * TL_ROWS + lane];
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk++) xn[kk] = lds[(xs + C0 + kk) * TL_ROWS + lane];
    // rows[kk] = the 16 x 32 strip [M_lo | M_hi] of the twiddle matrix for these 16 output planes; both 16 x 16 halves are
    // matrices of GF(2^16) elements: kara_acc
    kara_acc<16, 0, 0>(xn, y, rows);
    kara_acc<16, 0, 16>(xn, y + 16, rows);
#pragma unroll
    for (int kk = 0; kk < 16; kk++) yn[kk] = INV ? y[C0 + kk] : y[C0 + kk] ^ xn[kk];
}

template <bool INV>
__device__ __forceinline__ void low_word_levels(uint32_t *lds, const tile_desc &d, uint32_t base_w,
                                                const uint32_t *__restrict__ TW, uint32_t wave, uint32_t lane) {
    const uint32_t nl = d.nb - 5;
    if (nl == 0) return;
    const uint32_t items = 1u << nl;  // 2^(nl-1) butterflies x 2 halves, <= 16
    for (uint32_t qi = 0; qi < nl; qi++) {
        const uint32_t b = INV ? nl - 1 - qi : qi;
        uint32_t xn[16], yn[16];
        uint32_t xs = 0, ys = 0;
        const bool act = wave < items;
        const uint32_t c0 = (wave & 1u) << 4;
        if (act) {
            const uint32_t bf = wave >> 1;
            const uint32_t lvx = ((bf >> b) << (b + 1)) | (bf & ((1u << b) - 1u));
            xs = lvx << 5;
            ys = xs + (32u << b);
            const uint32_t w = base_w | tile_word(d, xs);
            const uint32_t r = d.m - 4 - nl + b;  // the tile holds the lowest nl word levels (17..20 at m = 25)
            const uint32_t q = ((1u << (d.m - 5)) | (w & ~d.pmw)) >> (d.m - r - 5);
            if (c0)
                low_word_item<INV, 16>(lds, xs, ys, TW, q, lane, xn, yn);
            else
                low_word_item<INV, 0>(lds, xs, ys, TW, q, lane, xn, yn);
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int kk = 0; kk < 16; kk++) {
                lds[(xs + c0 + kk) * TL_ROWS + lane] = xn[kk];
                lds[(ys + c0 + kk) * TL_ROWS + lane] = yn[kk];
            }
        }
        __syncthreads();
    }
}

// the intra-word part of the low kernel: wave w takes value word w of the tile (see k_fft_low)
template <int MODE, int NVW>
__device__ __forceinline__ void low_intra(uint32_t *lds, const uint32_t *__restrict__ LM, const uint32_t *__restrict__ Amat,
                                          uint32_t o, uint32_t nvw, uint32_t wave, uint32_t lane) {
    for (uint32_t vi = wave; vi < nvw; vi += NVW) {
        uint32_t *p = lds + (vi << 5) * TL_ROWS + lane;
        const uint64_t vw = (uint64_t)o * nvw + vi;  // wave-uniform
        const uint32_t *L = LM + vw * (5 * LOW_BLK_WORDS);
        const uint32_t *A = Amat + vw * AMAT_WORDS;
        // While W is in registers the 8 KiB of this value word in LDS are free and private to the wave: a ring of
        // LOW_RING mask blocks of at most 960 B, filled LOW_RING - 1 blocks ahead with plain vector loads (16 B per lane and
        // block). Blocks in the order they are consumed: forward levels
        // 4..0, [the two blocks of Amat, inverse levels 0..4].
        constexpr int NBLK = MODE == 1 ? 12 : 5, AH = LOW_RING - 1 < NBLK ? LOW_RING - 1 : NBLK;
        uint4 *R = (uint4 *)(lds + (vi << 5) * TL_ROWS);
        const uint32_t li = lane < LOW_SLOT ? lane : LOW_SLOT - 1;  // a block is at most 60 uint4: one per lane
        auto src = [&](int b) {
            return (const uint4 *)(lowblk_is_amat(b) ? A + (b - 5) * AMAT_B0_WORDS : L + lowblk_level(b) * LOW_BLK_WORDS);
        };
        uint4 s0[AH];
#pragma unroll
        for (int b = 0; b < AH; b++) s0[b] = src(b)[li];
        uint32_t W[32];
#pragma unroll
        for (int k = 0; k < 32; k++) W[k] = p[k * TL_ROWS];
#pragma unroll
        for (int b = 0; b < AH; b++) R[b * LOW_SLOT + li] = s0[b];
        uint32_t t1[16], t2[8];
#pragma unroll
        for (int b = 0; b < NBLK; b++) {
            uint4 n0;
            if (b + AH < NBLK) n0 = src(b + AH)[li];
            const uint4 *M = R + (b % LOW_RING) * LOW_SLOT;
            switch (b) {
                case 0: low_level_lds<4, false>(W, M); break;
                case 1: low_level_lds<3, false>(W, M); break;
                case 2: low_level_lds<2, false>(W, M); break;
                case 3: low_level_lds<1, false>(W, M); break;
                case 4: low_level_lds<0, false>(W, M); break;
                case 5: amat_block0(W, M, t1, t2); break;
                case 6: amat_block1(W, M, t1, t2); break;
                case 7: low_level_lds<0, true>(W, M); break;
                case 8: low_level_lds<1, true>(W, M); break;
                case 9: low_level_lds<2, true>(W, M); break;
                case 10: low_level_lds<3, true>(W, M); break;
                default: low_level_lds<4, true>(W, M); break;
            }
            if (b + AH < NBLK) R[((b + AH) % LOW_RING) * LOW_SLOT + li] = n0;
        }
#pragma unroll
        for (int k = 0; k < 32; k++) p[k * TL_ROWS] = W[k];
    }
}

// K5: the lowest word levels (those of the tile) + the intra-word levels, forward; MODE 1 continues with the
// pointwise multiply by FFT(a) and the same levels inverse. MODE 0 is used once, for FFT(a) itself.
// Tile = 5 plane bits + the lowest (up to 4) word bits: up to 16 value words x 32 planes x 64 rows = 128 KiB.
// NVW = value words per tile = 2^(d.nb - 5) = waves per workgroup (one value word per wave in the intra-word part, one
// work item per wave in the word levels): 16 for m >= 25, fewer for the small transforms (m = 21: a single wave).
template <int MODE, int NVW>
__global__ void __launch_bounds__(64 * NVW)
    k_fft_low(uint32_t *__restrict__ Z, tile_desc d, const uint32_t *__restrict__ TW, const uint32_t *__restrict__ LM,
              const uint32_t *__restrict__ Amat) {
    constexpr int NBK = NVW == 16 ? 9 : NVW == 8 ? 8 : NVW == 4 ? 7 : NVW == 2 ? 6 : 5;
    __shared__ uint32_t lds[32 * NVW * TL_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the two row halves of a tile read the same 11 KiB of mask blocks per value word (12 blocks of up to 960 B): they go to workgroups b and b + 8, which
    // run on the same XCD (workgroup b -> XCD b % 8) next to each other in launch order, so the second finds them in
    // that XCD's L2 (with rh = b & 1 they sat on two XCDs and every mask line came from HBM twice)
    uint32_t rh = blockIdx.x & 1, o = blockIdx.x >> 1;
    if (d.rhalves == 1) {
        rh = 0, o = blockIdx.x;
    } else if (gridDim.x % 16 == 0) {
        rh = (blockIdx.x >> 3) & 1;
        o = ((blockIdx.x >> 4) << 3) | (blockIdx.x & 7);
    }
    const uint32_t base_w = tile_base(d, o);
    const uint32_t nvw = (QC_EXP & 2) ? 0 : 1u << (d.nb - 5);  // value words in the tile
    tile_load<NBK, NVW>(lds, Z, d, base_w, rh, wave, lane);
    __syncthreads();
    if (!(QC_EXP & 1)) low_word_levels<false>(lds, d, base_w, TW, wave, lane);
    low_intra<MODE, NVW>(lds, LM, Amat, o, nvw, wave, lane);
    __syncthreads();
    if (MODE == 1 && !(QC_EXP & 1)) low_word_levels<true>(lds, d, base_w, TW, wave, lane);
    tile_store<NBK, NVW>(lds, Z, d, base_w, rh, wave, lane);
}

// The same kernel, persistent, for the full-size tile (MODE 1, 16 value words, one workgroup per CU): the next tile is
// fetched into registers (32 per thread) when the intra-word part, which needs the registers, is over, travels during
// the inverse word levels, and is put into LDS after this tile's stores were issued -- load and store phases (0.23 of
// 1.4 ms per launch) no longer leave the CU idle. idx -> (rh, o) as above with idx in place of blockIdx.x: workgroups
// b and b ^ 8 (same XCD) walk the two row halves of the same tiles side by side.
__global__ void __launch_bounds__(1024)
    k_fft_low_p(uint32_t *__restrict__ Z, tile_desc d, const uint32_t *__restrict__ TW, const uint32_t *__restrict__ LM,
                const uint32_t *__restrict__ Amat, uint32_t ntiles) {
    __shared__ uint32_t lds[32 * 16 * TL_ROWS];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto map = [&](uint32_t idx, uint32_t &rh, uint32_t &o) {
        rh = idx & 1, o = idx >> 1;
        if (d.rhalves == 1) {
            rh = 0, o = idx;
        } else if (ntiles % 16 == 0) {
            rh = (idx >> 3) & 1;
            o = ((idx >> 4) << 3) | (idx & 7);
        }
    };
    uint32_t idx = blockIdx.x;
    if (idx >= ntiles) return;
    uint32_t rh, o;
    map(idx, rh, o);
    typename tile_regs<32>::type v = tile_fetch<9>(Z, d, tile_base(d, o), rh, wave, lane);
#pragma unroll 1
    for (; idx < ntiles; idx += gridDim.x) {
        map(idx, rh, o);
        const uint32_t base_w = tile_base(d, o);
        tile_put<9>(lds, v, d, wave, lane);
        __syncthreads();
        low_word_levels<false>(lds, d, base_w, TW, wave, lane);
        low_intra<1, 16>(lds, LM, Amat, o, 16u, wave, lane);
        __syncthreads();
        {
            const uint32_t nidx = idx + gridDim.x < ntiles ? idx + gridDim.x : idx;
            uint32_t nrh, no;
            map(nidx, nrh, no);
            v = tile_fetch<9>(Z, d, tile_base(d, no), nrh, wave, lane);
        }
        low_word_levels<true>(lds, d, base_w, TW, wave, lane);
        tile_store<9, 16>(lds, Z, d, base_w, rh, wave, lane);
        __syncthreads();
    }
}

template <int MODE>
static void launch_low(sot_ctx *c, uint32_t *Z, const tile_desc &d) {
    const dim3 g((1u << (d.m - 5 - d.nb)) * d.rhalves);
    switch (d.nb) {
        case 5: hipLaunchKernelGGL((k_fft_low<MODE, 1>), g, dim3(64), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur); break;
        case 6: hipLaunchKernelGGL((k_fft_low<MODE, 2>), g, dim3(128), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur); break;
        case 7: hipLaunchKernelGGL((k_fft_low<MODE, 4>), g, dim3(256), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur); break;
        case 8: hipLaunchKernelGGL((k_fft_low<MODE, 8>), g, dim3(512), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur); break;
        default:
            if (MODE == 1 && !QC_EXP && g.x > c->num_cus && !(c->opts & SOT_OPT_NO_TOP)) {
                hipLaunchKernelGGL(k_fft_low_p, dim3(c->num_cus), dim3(1024), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur, g.x);
                break;
            }
            hipLaunchKernelGGL((k_fft_low<MODE, 16>), g, dim3(1024), 0, c->stream, Z, d, c->TW, c->LM, c->Amat_cur);
            break;
    }
}

// LM[(value word * 5 + i) * 384 + t]: the packed twiddle masks of intra-word level i (see low_level_lds). Bit p of a
// mask belongs to the pair whose x position is p with bit i cleared; R = row of that pair's twiddle matrix.
//   t < 256:  j = t / 16, k = t % 16: bit j of row k at x positions (A), bit j + 16 at y positions (X)
//   t >= 256: j' = (t - 256) / 16, k: bit 16 + j' of row k + 16 at x positions, bit 24 + j' at y positions (A + B)
__global__ void k_build_lowmasks(tile_desc d, uint64_t nvals, const uint32_t *__restrict__ TW, uint32_t *__restrict__ LM) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvals * 5 * LOW_BLK_PLAIN) return;  // one thread per word of the plain layout; unread words are dropped
    const uint32_t t = (uint32_t)(idx % LOW_BLK_PLAIN);
    const uint32_t li = (uint32_t)((idx / LOW_BLK_PLAIN) % 5);
    const uint32_t vw = (uint32_t)(idx / (5 * LOW_BLK_PLAIN));
    const int cu = d_LOWCMAP[0][t >> 2];
    if (cu < 0) return;
    const bool set2 = t >= 256;
    const uint32_t j = set2 ? (t - 256) >> 4 : t >> 4, k = (t & 15) + (set2 ? 16 : 0);
    const uint32_t nvw = 1u << (d.nb - 5);
    const uint32_t w = tile_base(d, vw / nvw) | tile_word(d, (vw % nvw) << 5);
    const uint32_t q = ((1u << d.m) | ((w & ~d.pmw) << 5)) >> li;
    uint32_t rw[1];
    tw_rows<1>(TW, q, k, rw);
    uint32_t mask = 0;
    for (uint32_t p = 0; p < 32; p++) {
        const uint32_t e = (p & ~(1u << li)) >> li;  // per-position twiddle index of the pair's x position
        const uint32_t R = rw[0] ^ TW[(e << 5) + k];
        const uint32_t ypos = (p >> li) & 1u;
        const uint32_t jj = set2 ? 16 + j + 8 * ypos : j + 16 * ypos;
        mask |= ((R >> jj) & 1u) << p;
    }
    LM[((uint64_t)vw * 5 + li) * LOW_BLK_WORDS + (uint32_t)cu * 4 + (t & 3u)] = mask;
}

// Amat[(o * nvw + vi) * 768 + t] bit p = entry (k, j) of the matrix 'multiply by the FFT(a) value at position p of value
// word (o, vi)' (bit k of value (x) 2^j; row 0 of Z): t < 256: (k, j) = (t % 16, t / 16); 256 <= t < 512:
// (t % 16, 16 + (t - 256) / 16); t >= 512: (16 + t % 16, 16 + (t - 512) / 16). See amat_block0/1.
__global__ void k_build_amat(const uint32_t *__restrict__ Z, tile_desc d, uint64_t nvals, uint32_t *__restrict__ Amat) {
    uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nvals * AMAT_PLAIN) return;  // one thread per word of the plain layout (two blocks of 96 uint4)
    const uint32_t vw = (uint32_t)(idx / AMAT_PLAIN), t = (uint32_t)(idx % AMAT_PLAIN);
    const uint32_t blk = t >= LOW_BLK_PLAIN, tb = t - blk * LOW_BLK_PLAIN;
    const int cu = d_LOWCMAP[1 + blk][tb >> 2];
    if (cu < 0) return;
    const uint32_t set = t >> 8;
    const uint32_t j = ((t & 255u) >> 4) + (set ? 16 : 0), k = (t & 15u) + (set == 2 ? 16 : 0);
    const uint32_t nvw = 1u << (d.nb - 5);
    const uint32_t w = tile_base(d, vw / nvw) | tile_word(d, (vw % nvw) << 5);
    uint32_t mm = 0;
    for (uint32_t tt = 0; tt < 32; tt++)
        if ((d_RTAB[tt * 32 + k] >> j) & 1) mm ^= Z[(uint64_t)(w | tile_word(d, tt)) * ROWS];
    Amat[(uint64_t)vw * AMAT_WORDS + blk * AMAT_B0_WORDS + (uint32_t)cu * 4 + (tb & 3u)] = mm;
}

// ------------------------------------------------------------------------------------------------
// K10: fold mod x^P - 1, add V0, add the tail terms:
//   B[k] = V0[k] ^ V0[k+P] ^ T[k] ^ T[k+P] ^ sum_{i: a_hi[i]} V1'[(k - S - i) mod P]
//                                         ^ sum_{j < tail} a_lo[(k - S - j) mod P] V1'[S + j]
// V0 = V[0 .. nb128), V1' = (V1 mod x^P - 1). a2 = a' repeated twice as a bit vector.
__device__ __forceinline__ blk_t ld_v1p(const blk_t *V, const uint32_t *sb, uint32_t sbpos, uint64_t v1_len, uint64_t P,
                                        uint64_t c, uint32_t lw) {
    blk_t v = ld_leaf(V, v1_len + c, sb, sbpos, lw);  // V1 = V + nb128, v1_len == nb128
    if (c + P < v1_len) {
        blk_t u = ld_leaf(V, v1_len + c + P, sb, sbpos, lw);
        v.x ^= u.x;
        v.y ^= u.y;
        v.z ^= u.z;
        v.w ^= u.w;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// K10 fused with K9: product rows -> Blocks, fold mod x^P - 1, add V0 and the tail terms, in one pass (no T):
//   B[k] = V0[k] ^ V0[k+P] ^ C[k] ^ C[k+P] ^ sum_{i: a_hi[i]} V1'[(q - i) mod P] ^ sum_{j < tail} a_lo[(q - j) mod P] V1'[S + j]
// with C = a_lo * V1'_lo (bit rows in Z, C has fewer than 2 min(P, S) bits) and q = (k - S) mod P.
// One workgroup = FN_W words = 512 consecutive k. The row words of C[k] and C[k+P] are combined before the 32x32
// transposes (P is odd: a funnel shift of two words); the tail sums read a window of V1' and of a' staged in LDS
// (index x = (k - k0) + (tail - 1 - i) for both).
#define FN_W 16
#define FN_K (FN_W * 32)
#define FN_TAILMAX 256
static inline size_t fn_dyn_lds(uint32_t tail) {
    return tail ? (size_t)(FN_K + 2 * tail) * 16 + ((FN_K + tail + 15) & ~15u) : 0;
}
template <bool FULL>  // as in k_qc_load
__global__ void __launch_bounds__(256)
    k_qc_finish(const uint32_t *__restrict__ Z, uint64_t nw, const blk_t *__restrict__ V, const uint32_t *__restrict__ sb,
                uint32_t sbpos, uint64_t nb128, uint64_t P, uint64_t S, uint32_t tail, uint64_t n, const uint32_t *__restrict__ a2,
                const uint32_t *__restrict__ a_hi, blk_t *__restrict__ out, uint32_t rows_active_, uint32_t out_words_,
                const blk_t *__restrict__ V0, const uint32_t *__restrict__ sb0) {
    // V / sb: the leaves re-based so that the multiplied segment b_s is "V1" (Blocks [nb128, 2 nb128)): the tail terms read
    // it. V0 / sb0: the leaves themselves for the b0 terms of the FIRST product (s = 1); nullptr for s >= 2 (scaler > 2),
    // whose product is XORed into what `out` already holds (quasi_cyclic_encode.rs:265-277: c_poly.add_eq).
    const uint32_t rows_active = FULL ? ROWS : rows_active_, out_words = FULL ? 4u : out_words_;
    __shared__ uint32_t tio[FN_W * 132];  // rows in: [w][row] (stride 129); Blocks out: [w][p][q] (stride 132)
    // sized by the launch (fn_dyn_lds): the kernel is bound by how many workgroups fit a CU, and tail << FN_TAILMAX
    extern __shared__ blk_t fn_dyn[];
    blk_t *win = fn_dyn;                                    // [FN_K + tail]  V1'[(c0 + x) mod P]
    blk_t *hi = fn_dyn + FN_K + tail;                       // [tail]         V1'[S + j]
    uint8_t *abit = (uint8_t *)(fn_dyn + FN_K + 2 * tail);  // [FN_K + tail]  0xff * a_lo[(c0 + x) mod P]
    const uint64_t k0 = (uint64_t)blockIdx.x * FN_K;
    const uint64_t w0 = k0 >> 5;
    const uint64_t wsh = P >> 5;
    const uint32_t sh = (uint32_t)(P & 31);
    for (uint32_t i = threadIdx.x; i < FN_W * ROWS; i += 256) {
        const uint64_t w = w0 + (i >> 7);
        const uint32_t row = i & 127;
        // rows >= rows_active are zero by linearity and were not computed (sharded contexts): never read them
        const bool live = row < rows_active;
        uint32_t v = 0;
        if (w < nw && live) v = Z[w * ROWS + row];
        const uint64_t w1 = w + wsh;
        if (w1 < nw && live) {
            uint32_t lo = Z[w1 * ROWS + row];
            uint32_t hi1 = (sh && w1 + 1 < nw) ? Z[(w1 + 1) * ROWS + row] : 0u;
            v ^= sh ? ((lo >> sh) | (hi1 << (32 - sh))) : lo;
        }
        tio[(i >> 7) * 129 + row] = v;
    }
    if (tail && !(QC_EXP & 4)) {
        const uint64_t q0 = k0 >= S ? k0 - S : k0 + P - S;  // k0 < n <= P
        const uint64_t c0 = (q0 + P - (tail - 1)) % P;
        for (uint32_t x = threadIdx.x; x < FN_K + tail - 1; x += 256) {
            uint64_t cc = c0 + x;
            if (cc >= P) cc -= P;
            win[x] = ld_v1p(V, sb, sbpos, nb128, P, cc, out_words);
            abit[x] = cc < S ? (uint8_t)(0u - ((a2[cc >> 5] >> (cc & 31)) & 1u)) : (uint8_t)0;
        }
        for (uint32_t j = threadIdx.x; j < tail; j += 256) hi[j] = ld_v1p(V, sb, sbpos, nb128, P, S + j, out_words);
    }
    __syncthreads();
    {
        // 32x32 bit transposes through registers, in place (128 of the 256 threads: word wl, quarter q of the rows)
        const uint32_t wl = threadIdx.x >> 2, q = threadIdx.x & 3;
        const bool act = threadIdx.x < FN_W * 4 && !(QC_EXP & 8);
        uint32_t a[32];
        if (act) {
            if (32 * q < rows_active) {
#pragma unroll
                for (int b = 0; b < 32; b++) a[b] = tio[wl * 129 + 32 * q + b];
                transpose32(a);
            } else {  // a quarter of dead rows: word q of every Block is zero
#pragma unroll
                for (int b = 0; b < 32; b++) a[b] = 0;
            }
        }
        __syncthreads();
        if (act) {
#pragma unroll
            for (int p = 0; p < 32; p++) tio[wl * 132 + p * 4 + q] = a[p];
        }
    }
    __syncthreads();
    for (uint32_t kl = threadIdx.x; kl < FN_K; kl += 256) {
        const uint64_t k = k0 + kl;
        if (k >= n) break;
        const uint32_t *sp = tio + (kl >> 5) * 132 + (kl & 31) * 4;
        blk_t r = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        blk_t u;
        if (V0) {
            u = ld_leaf(V0, k, sb0, sbpos, out_words);
            r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
            if (k + P < nb128) {
                u = ld_leaf(V0, k + P, sb0, sbpos, out_words);
                r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
            }
        } else {
            u = ld_leaf(out, k, nullptr, 0, out_words);  // packed like the leaves
            r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
        }
        if (tail && !(QC_EXP & 4)) {
            // branch-free in the lanes: a wave-uniform test of a_hi[i], a per-lane byte mask for a_lo
            const uint32_t xb = kl + tail - 1;  // x = xb - i
            for (uint32_t i = 0; i < tail; i++) {
                if ((a_hi[i >> 5] >> (i & 31)) & 1u) {  // wave-uniform
                    u = win[xb - i];
                    r.x ^= u.x, r.y ^= u.y, r.z ^= u.z, r.w ^= u.w;
                }
                const uint32_t m = (uint32_t)(int32_t)(int8_t)abit[xb - i];
                u = hi[i];
                r.x = XAND(r.x, u.x, m), r.y = XAND(r.y, u.y, m), r.z = XAND(r.z, u.z, m), r.w = XAND(r.w, u.w, m);
            }
        }
        // out_words < 4: only the low 64 / 32 bits of a leaf and of an output hold data (slot Blocks of a sharded context of
        // 4 / 8 ranks); both are stored packed (ld_leaf), and the packed outputs are what travels in exchange #2
        if (out_words == 4)
            out[k] = r;
        else if (out_words == 2)
            ((uint2 *)out)[k] = make_uint2(r.x, r.y);
        else
            ((uint32_t *)out)[k] = r.x;
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

// monomial <-> novel basis conversion of the 2^ell_elems low coefficients of every row, on bufs[cur].
// Phase list (gf2fft_model.conversion_phases) = [Taylor phases at x^(2^16)+x] [whole-unit phases] [phases local to a
// 2^16 unit]; the last two groups commute. Taylor phases: k_conv_taylor (groups A and B, out of place), whole-unit
// phases: inside the top butterfly pass when `units_in_top`, local phases: k_conv_tile (in place). Anything the
// fused kernels do not cover goes through the generic one-phase kernel. Returns the buffer index of the result.
// group A restricted to the phases pw = 2^pb_lo .. 2^pb_hi bits (5 <= pb_lo <= pb_hi <= R - 1) of an array of
// 2^(R+16) bits per row. Its largest chunk spans 2^(pb_hi-4) units of 2^21 bits (one group); every shift is a multiple
// of 2^(pb_lo-5) words = the column stride, so the halo is 2^(pb_hi-pb_lo+1) columns whatever R is.
static uint32_t taylor_nact(uint32_t lrq, uint32_t rows_active) {
    const uint32_t per = 4u << lrq, all = 32u >> lrq;
    uint32_t need = (rows_active + per - 1) / per, n = 1;
    while (n < need) n <<= 1;
    return n < all ? n : all;
}
static bool make_taylor_a(uint32_t R, uint32_t pb_hi, uint32_t pb_lo, taylor_desc &t, unsigned &grid, uint32_t &ncol,
                          uint32_t rows_active) {
    memset(&t, 0, sizeof t);
    if (pb_lo < 5 || pb_hi < pb_lo || pb_hi + 1 > R) return false;
    int lrq = 3;
    t.lus = 21;
    t.lnu = pb_hi - 4;
    t.lg = pb_lo - 5;
    t.hw = 1u << (pb_hi - pb_lo + 1);
    int lnc = 10 - (int)t.lnu;
    while (lnc < 0 || (1u << lnc) < 2 * t.hw) {
        lrq--;
        lnc++;
    }
    if (lrq < 0 || t.lus - 5 - t.lg < (uint32_t)lnc) return false;
    t.lnc = (uint32_t)lnc;
    for (int pb = (int)pb_hi; pb >= (int)pb_lo; pb--) {
        t.lchunk[t.nph] = (uint8_t)(pb + 17);
        t.lpw[t.nph] = (uint8_t)pb;
        t.nph++;
    }
    t.lrq = (uint32_t)lrq;
    const uint32_t UW = 1u << (t.lus - 5 - t.lg), W = (1u << t.lnc) - t.hw;
    ncol = (UW + W - 1) / W;
    const uint32_t groups = 1u << (R - 5 - t.lnu);
    t.nact = taylor_nact(t.lrq, rows_active);
    grid = (groups * ncol * t.nact) << t.lg;
    return true;
}

static bool make_taylor(uint32_t R, bool groupA, taylor_desc &t, unsigned &grid, uint32_t &ncol, uint32_t rows_active) {
    memset(&t, 0, sizeof t);
    int lrq = 3;
    if (!groupA) {
        t.lus = 16;
        t.lnu = R < 5 ? R : 5;
        t.hw = 1;
        t.lnc = 10 - t.lnu;
        for (int pb = (int)t.lnu - 1; pb >= 0; pb--) {
            t.lchunk[t.nph] = (uint8_t)(pb + 17);
            t.lpw[t.nph] = (uint8_t)pb;
            t.nph++;
        }
    } else {
        if (R <= 5) return false;
        t.lus = 21;
        t.lnu = R - 5;
        t.hw = 1u << t.lnu;
        int lnc = 10 - (int)t.lnu;
        while (lnc < 0 || (1u << lnc) < 2 * t.hw) {
            lrq--;
            lnc++;
        }
        if (lrq < 0) return false;
        t.lnc = (uint32_t)lnc;
        for (int pb = (int)R - 1; pb >= 5; pb--) {
            t.lchunk[t.nph] = (uint8_t)(pb + 17);
            t.lpw[t.nph] = (uint8_t)pb;
            t.nph++;
        }
    }
    t.lrq = (uint32_t)lrq;
    const uint32_t UW = 1u << (t.lus - 5), W = (1u << t.lnc) - t.hw;
    ncol = (UW + W - 1) / W;
    const uint32_t groups = 1u << (R - (groupA ? R : t.lnu));
    t.nact = taylor_nact(t.lrq, rows_active);
    grid = groups * ncol * t.nact;
    return true;
}

// a zeroed work counter for one launch of a kernel that draws its tiles (k_mid): slots of c->d_ctr are handed out in turn
// and the array is cleared, in stream order, when all have been used
static uint32_t *qc_next_counter(sot_ctx *c, uint32_t n = 1) {
    if (c->ctr_next + n > QC_NCTR) {
        (void)hipMemsetAsync(c->d_ctr, 0, QC_NCTR * sizeof(uint32_t), c->stream);
        c->ctr_next = 0;
    }
    c->ctr_next += n;
    return c->d_ctr + c->ctr_next - n;
}
static int run_conversion(sot_ctx *c, uint32_t *bufs[2], int cur, uint32_t ell_elems, bool inverse, bool units_in_top,
                          const move_list *sub_fwd, const move_list *sub_inv) {
    const uint64_t nwords = 1ull << (ell_elems - 5);
    std::vector<phase_t> ph;
    conv_rec(ph, ell_elems, 1);
    const uint32_t tb = ell_elems < 16 ? ell_elems : 16;
    std::vector<phase_t> taylor, units;
    conv_phases small;
    small.n = 0;
    for (auto &p : ph) {
        if (p.chunk > (1ull << tb)) {
            (p.pw < (1ull << tb) ? taylor : units).push_back(p);
        } else {
            small.lchunk[small.n] = (uint8_t)ilog2(p.chunk);
            small.lpw[small.n] = (uint8_t)ilog2(p.pw);
            small.n++;
        }
    }
    const unsigned grid = gridn(nwords * ROWS, 256, 1u << 20);
    const unsigned tgrid = (unsigned)((nwords >> (tb - 5)) * 8);
    // row slices that hold data (sharded contexts of 4 or more ranks: qc_set_rows_active); powers of two
    const uint32_t ra = c->rows_active;
    const uint32_t nact16 = taylor_nact(QC_CU_RQ == 4 ? 2 : 1, ra);  // slices of 4 QC_CU_RQ rows
    const unsigned ugrid = (unsigned)((nwords >> (tb - 5)) * nact16);
    const unsigned ucap = (QC_CU_RQ == 4 ? 1 : 2) * c->num_cus;  // resident workgroups
    const int kid = inverse ? KID_CONV_INV : KID_CONV_FWD;
    const int kid_t = inverse ? KID_CONV_INV_TILE : KID_CONV_FWD_TILE;
    const uint32_t R = ell_elems > 16 ? ell_elems - 16 : 0;
    taylor_desc tA, tB;
    unsigned gA = 0, gB = 0;
    uint32_t ncA = 0, ncB = 0;
    const bool fused = R >= 1 && !(c->opts & SOT_OPT_NO_TOP);
    const bool haveB = fused && make_taylor(R, false, tB, gB, ncB, ra);
    const bool needA = R > 5;
    bool haveA = needA && haveB && make_taylor(R, true, tA, gA, ncA, ra);
    // R >= 11: one group-A tile would be half halo with 16-byte row spans (4.8 ms for one launch at R = 11, n = 2^26).
    // Split: A2 = the phases pw >= 2^9 bits on columns 16 words apart (halo 4 columns), then A = the R = 9 tile
    // geometry (pw = 2^8 .. 2^5) on groups of 16 units: 1.1 ms each. (At R = 10 the split measures the same as the
    // single tile with 32-byte spans, so it keeps one launch.)
    taylor_desc tA2;
    unsigned gA2 = 0;
    uint32_t ncA2 = 0;
    bool haveA2 = false;
    if (haveA && R >= 11 && !(c->opts & SOT_OPT_NO_ASPLIT)) {
        taylor_desc lo, hi;
        unsigned glo = 0, ghi = 0;
        uint32_t nlo = 0, nhi = 0;
        if (make_taylor_a(R, 8, 5, lo, glo, nlo, ra) && make_taylor_a(R, R - 1, 9, hi, ghi, nhi, ra)) {
            tA = lo, gA = glo, ncA = nlo;
            tA2 = hi, gA2 = ghi, ncA2 = nhi;
            haveA2 = true;
        }
    }
    const bool fusedT = haveB && (!needA || haveA);
    auto generic = [&](const std::vector<phase_t> &v) {
        for (size_t q = 0; q < v.size(); q++) {
            const phase_t &p = inverse ? v[v.size() - 1 - q] : v[q];
            SOT_LAUNCH(c, kid, k_conv_phase<1>, dim3(grid), dim3(256), bufs[cur], bufs[cur ^ 1], nwords, p.chunk, p.pw,
                       p.half, inverse ? 1 : 0);
            cur ^= 1;
        }
    };
    auto run_taylor = [&](const taylor_desc &t, unsigned g, uint32_t nc) {
        SOT_LAUNCH(c, kid, k_conv_taylor, dim3(g < c->num_cus ? g : c->num_cus), dim3(CT_THREADS), bufs[cur], bufs[cur ^ 1], t,
                   nc, inverse ? 1 : 0, (uint32_t)g);
        cur ^= 1;
    };
    // whole-unit phases outside the top pass: one in-place tile kernel when all units of an offset fit LDS
    const uint32_t nub = ell_elems > 16 ? ell_elems - 16 : 0;
    auto whole_units = [&]() {
        if (units.empty()) return;
        if (nub < 8 || nub > 11 || units.size() > 32 || (c->opts & SOT_OPT_NO_WU)) {
            generic(units);
            return;
        }
        wu_phases w;
        memset(&w, 0, sizeof w);
        w.nub = nub;
        w.nact = taylor_nact(13 - nub, ra);
        for (auto &p : units) {
            w.lchunk[w.nph] = (uint8_t)(ilog2(p.chunk) - 16);
            w.lpw[w.nph] = (uint8_t)(ilog2(p.pw) - 16);
            w.nph++;
        }
        SOT_LAUNCH(c, kid, k_conv_wholeunits, dim3(c->num_cus), dim3(CT_THREADS), bufs[cur], w, inverse ? 1 : 0);
    };
    if (!inverse) {
        if (fusedT) {
            if (haveA2) run_taylor(tA2, gA2, ncA2);
            if (haveA) run_taylor(tA, gA, ncA);
            run_taylor(tB, gB, ncB);
        } else {
            generic(taylor);
        }
        if (!units_in_top) whole_units();
        if (fused && tb == 16)
            SOT_LAUNCH(c, kid_t, (k_conv_unit<false, QC_CU_RQ>), dim3(ugrid < ucap ? ugrid : ucap), dim3(256 * QC_CU_RQ), bufs[cur], *sub_fwd, ugrid, nact16);
        else if (small.n)
            SOT_LAUNCH(c, kid_t, k_conv_tile<1>, dim3(tgrid), dim3(CT_THREADS), bufs[cur], tb, small, 0);
    } else {
        if (fused && tb == 16)
            SOT_LAUNCH(c, kid_t, (k_conv_unit<true, QC_CU_RQ>), dim3(ugrid < ucap ? ugrid : ucap), dim3(256 * QC_CU_RQ), bufs[cur], *sub_inv, ugrid, nact16);
        else if (small.n)
            SOT_LAUNCH(c, kid_t, k_conv_tile<1>, dim3(tgrid), dim3(CT_THREADS), bufs[cur], tb, small, 1);
        if (!units_in_top) whole_units();
        if (fusedT) {
            run_taylor(tB, gB, ncB);
            if (haveA) run_taylor(tA, gA, ncA);
            if (haveA2) run_taylor(tA2, gA2, ncA2);
        } else {
            generic(taylor);
        }
    }
    return cur;
}

// ---- tile descriptors of the butterfly passes (see fafft_model.py: level r <-> word-index bit m - r - 5)
static inline bool is_pruned(uint32_t r) { return r == 1 || r == 2 || r == 4 || r == 8 || r == 16; }

static tile_desc make_desc(uint32_t m, const std::vector<uint32_t> &plane_r, const std::vector<uint32_t> &level_r,
                           const std::vector<uint32_t> &extra_wbits = {}) {
    // plane_r: pruned levels whose word bit is (or becomes, inside this pass) a plane bit, in plane-bit order;
    // level_r: levels run by this pass, in forward order (ascending r)
    tile_desc d;
    memset(&d, 0, sizeof d);
    d.m = m;
    d.zero_bit = 32;
    d.rhalves = 2;
    auto wb = [&](uint32_t r) { return m - r - 5; };
    for (uint32_t r : {1u, 2u, 4u, 8u, 16u}) d.pmw |= 1u << wb(r);
    std::vector<uint32_t> bits_r = plane_r;
    for (uint32_t r : level_r)
        if (!is_pruned(r)) bits_r.push_back(r);
    d.nb = (uint32_t)(bits_r.size() + extra_wbits.size());
    for (uint32_t j = 0; j < d.nb; j++) {
        const uint32_t pos = j < bits_r.size() ? wb(bits_r[j]) : extra_wbits[j - bits_r.size()];
        d.posq |= (uint64_t)pos << (5 * j);
        d.tmask |= 1u << pos;
    }
    auto tbit_of = [&](uint32_t r) {
        for (uint32_t j = 0; j < d.nb; j++)
            if (bits_r[j] == r) return j;
        return 0u;
    };
    d.nlev = 0;
    for (uint32_t r : level_r) {
        if (r <= 2) continue;  // tau_lo = 0: these two pruned levels only relabel bits as planes
        uint32_t np = 0;
        for (uint32_t p : {1u, 2u, 4u, 8u, 16u})
            if (p < r) np++;
        d.lev[d.nlev++] = r | (tbit_of(r) << 8) | (np << 16) | ((is_pruned(r) ? 1u : 0u) << 24);
    }
    return d;
}

struct fft_plan {
    tile_desc top, mid1, mid2, low;
    uint32_t mid1_hi, mid2_hi;   // word bits (outside the tile) that the twiddles of the pass depend on
    bool units_in_top;           // the whole-unit conversion phases run inside the top pass
    unit_phases up_fwd, up_inv;  // for an operand (2^(m-1) coefficients) / for the product (2^m)
    unit_phases up_none;
    std::vector<uint32_t> list_fwd, list_inv;  // host copies of the move lists (uploaded once per context)
    move_list sub_fwd, sub_inv;                // sub-unit moves of k_conv_unit
    std::vector<uint32_t> sub_list_fwd, sub_list_inv;
    std::vector<tile_desc> w32;  // extra 32-plane word-level passes (m > 25), forward order
};

static fft_plan make_plan(uint32_t m, uint32_t opts) {
    fft_plan p;
    // unit-index bits = word bits 11 .. m-6; the top pass owns word bits m-13 .. m-6
    std::vector<uint32_t> extra;
    for (uint32_t b = 11; b + 13 < m; b++) extra.push_back(b);
    memset(&p.up_none, 0, sizeof p.up_none);
    p.up_fwd = p.up_inv = p.up_none;
    p.units_in_top = extra.size() <= 1 && !(opts & SOT_OPT_NO_TOP);
    if (!p.units_in_top) extra.clear();
    p.top = make_desc(m, {1, 2, 4, 8}, {1, 2, 3, 4, 5, 6, 7, 8}, extra);
    if (p.units_in_top) {
        // unit-index bit k (word bit 11 + k) -> tile bit; the remaining tile bits enumerate independent batches
        uint32_t ub[16] = {0}, bb[16] = {0}, nbb = 0;
        const uint32_t nub = m - 16;
        for (uint32_t j = 0; j < p.top.nb; j++) {
            const uint32_t pos = (uint32_t)(p.top.posq >> (5 * j)) & 31u;
            if (pos >= 11)
                ub[pos - 11] = j;
            else
                bb[nbb++] = j;
        }
        auto slot = [&](uint32_t u, uint32_t b) {
            uint32_t sl = 0;
            for (uint32_t k = 0; k < nub; k++) sl |= ((u >> k) & 1u) << ub[k];
            for (uint32_t k = 0; k < nbb; k++) sl |= ((b >> k) & 1u) << bb[k];
            return sl;
        };
        for (int inv = 0; inv < 2; inv++) {
            const uint32_t nub_act = inv ? m - 16 : m - 17;  // an operand only fills the lower half of the units
            std::vector<phase_t> ph;
            conv_rec(ph, nub_act, 1);
            std::vector<uint32_t> &list = inv ? p.list_inv : p.list_fwd;
            unit_phases &u = inv ? p.up_inv : p.up_fwd;
            for (size_t q = 0; q < ph.size(); q++) {
                const phase_t &f = inv ? ph[ph.size() - 1 - q] : ph[q];
                const uint32_t D = (uint32_t)(f.half - f.pw), nchunks = (1u << nub_act) / (uint32_t)f.chunk;
                // one step per phase: entry j < D carries the shift lo[pw + j] ^= H[j] and, for j < pw, the fold
                // H[j] ^= H[j + D] (tile_unit_phases)
                for (uint32_t cidx = 0; cidx < nchunks; cidx++)
                    for (uint32_t jj = 0; jj < D; jj++)
                        for (uint32_t b = 0; b < (1u << nbb); b++) {
                            const uint32_t base = cidx * (uint32_t)f.chunk;
                            const uint32_t h = base + (uint32_t)f.half + jj, lo = base + (uint32_t)f.pw + jj;
                            const bool fold = jj < (uint32_t)f.pw;
                            list.push_back(PH_ENTRY(slot(h, b), slot(lo, b), fold ? slot(h + D, b) : 0u, fold));
                        }
                u.step_end[u.nsteps++] = (uint32_t)list.size();
            }
            if (list.size() > TOP_MAX_MOVES) abort();  // 3328 at m = 25, the largest m with the units in the top pass
        }
    }
    // sub-unit moves of k_conv_unit: conversion of the 256 sub-unit indices of a unit
    memset(&p.sub_fwd, 0, sizeof p.sub_fwd);
    memset(&p.sub_inv, 0, sizeof p.sub_inv);
    for (int inv = 0; inv < 2; inv++) {
        std::vector<phase_t> ph;
        conv_rec(ph, 8, 1);
        std::vector<uint32_t> &list = inv ? p.sub_list_inv : p.sub_list_fwd;
        move_list &u = inv ? p.sub_inv : p.sub_fwd;
        for (size_t q = 0; q < ph.size(); q++) {
            const phase_t &f = inv ? ph[ph.size() - 1 - q] : ph[q];
            const uint32_t D = (uint32_t)(f.half - f.pw), nchunks = 256u / (uint32_t)f.chunk;
            for (uint32_t cidx = 0; cidx < nchunks; cidx++)
                for (uint32_t jj = 0; jj < D; jj++) {
                    const uint32_t base = cidx * (uint32_t)f.chunk;
                    const uint32_t h = base + (uint32_t)f.half + jj, lo = base + (uint32_t)f.pw + jj;
                    const bool fold = jj < (uint32_t)f.pw;
                    list.push_back(PH_ENTRY(h, lo, fold ? h + D : 0u, fold));
                }
            u.step_end[u.nsteps++] = (uint32_t)list.size();
        }
        if (list.size() > CU_MAX_MOVES || u.nsteps == 0) abort();  // a constant of the algorithm, not of the input
    }
    // 16-plane passes (k_mid): tile bits = 4 plane bits, then the 4 level bits in level order
    p.mid1 = make_desc(m, {1, 2, 4, 8}, {9, 10, 11, 12});
    p.mid2 = make_desc(m, {1, 2, 4, 8}, {13, 14, 15}, {m - 16 - 5});
    p.mid1_hi = p.mid2_hi = 0;
    for (uint32_t r : {3u, 5u, 6u, 7u}) p.mid1_hi |= 1u << (m - r - 5);
    p.mid2_hi = p.mid1_hi;
    for (uint32_t r : {9u, 10u, 11u, 12u}) p.mid2_hi |= 1u << (m - r - 5);
    // 32-plane word levels r = 17 .. m-5; the lowest (up to 4) belong to the fused low kernel
    const uint32_t last = m - 5;
    uint32_t nlow = last >= 17 ? last - 16 : 0;
    uint32_t in_low = nlow < 4 ? nlow : 4;
    uint32_t r = 17;
    while (last + 1 - r > in_low) {
        uint32_t cnt = last + 1 - r - in_low;
        if (cnt > 4) cnt = 4;
        std::vector<uint32_t> lv;
        for (uint32_t k = 0; k < cnt; k++) lv.push_back(r + k);
        // 5 plane bits + cnt level bits are only 2^(5+cnt) slots: fill the tile up to 2^7 slots with the lowest word
        // bits (independent batches below every level bit, so the twiddles do not depend on them): 8 work items per
        // level = one per wave of a row half in k_tile_fft_pair
        std::vector<uint32_t> batch;
        for (uint32_t b = 0; 5 + cnt + batch.size() < ((opts & SOT_OPT_NO_W32PAIR) ? 8u : 7u); b++) batch.push_back(b);
        p.w32.push_back(make_desc(m, {1, 2, 4, 8, 16}, lv, batch));
        r += cnt;
    }
    std::vector<uint32_t> lv;
    for (; r <= last; r++) lv.push_back(r);
    p.low = make_desc(m, {1, 2, 4, 8, 16}, lv);
    return p;
}

static inline unsigned tile_grid(const tile_desc &d) { return (unsigned)((1u << (d.m - 5 - d.nb)) * d.rhalves); }

static void launch_tile(sot_ctx *c, int kid, bool inv, const tile_desc &d_in, uint32_t *Z, const unit_phases &up_in) {
    tile_desc d = d_in;
    unit_phases up = up_in;
#if QC_EXP  // profiling experiments only (tools/exp.sh builds with -DQC_EXP=...): these SKIP computation, so the default
            // library must not contain them
    if (getenv("SOT_DEBUG_NOLEVELS")) d.nlev = 0;
    if (getenv("SOT_DEBUG_NOUNITS")) up.nsteps = 0;
    if (const char *e = getenv("SOT_DEBUG_POS")) {  // experiment: synthetic tile bit positions, no compute
        d.nlev = 0, up.nsteps = 0, d.tmask = 0, d.zero_bit = 32;
        int mode = atoi(e);
        d.posq = 0;
        for (uint32_t j = 0; j < d.nb; j++) {
            const uint32_t pos = mode == 0 ? j : mode == 1 ? (j < 4 ? j : j + 8) : (j < 4 ? j + 4 : j + 8);
            d.posq |= (uint64_t)pos << (5 * j);
            d.tmask |= 1u << pos;
        }
    }
#endif
    const dim3 g(tile_grid(d)), b(TL_THREADS);
    if (d.nb <= 8) {
        if (inv)
            SOT_LAUNCH(c, kid, (k_tile_fft<true, 8>), g, b, Z, d, c->TW, up);
        else
            SOT_LAUNCH(c, kid, (k_tile_fft<false, 8>), g, b, Z, d, c->TW, up);
    } else {
        if (inv)
            SOT_LAUNCH(c, kid, (k_tile_fft<true, 9>), g, b, Z, d, c->TW, up);
        else
            SOT_LAUNCH(c, kid, (k_tile_fft<false, 9>), g, b, Z, d, c->TW, up);
    }
}

static void launch_w32(sot_ctx *c, int kid, bool inv, const tile_desc &d, uint32_t *Z, const unit_phases &up_none) {
    bool all32 = d.nb == 7;
    for (uint32_t l = 0; l < d.nlev; l++)
        if (((d.lev[l] >> 16) & 255u) != 5 || (d.lev[l] >> 24)) all32 = false;
    if (!all32 || d.rhalves == 1) {  // the pair kernel exists to share scalar masks between the two row halves
        launch_tile(c, kid, inv, d, Z, up_none);
        return;
    }
    const dim3 g(tile_grid(d) / 2), b(W32_THREADS);
    if (inv)
        SOT_LAUNCH(c, kid, (k_tile_fft_pair<true>), g, b, Z, d, c->TW);
    else
        SOT_LAUNCH(c, kid, (k_tile_fft_pair<false>), g, b, Z, d, c->TW);
}

static void launch_top(sot_ctx *c, int kid, bool inv, const tile_desc &d, uint32_t *Z, const unit_phases &up) {
    const uint32_t nt = tile_grid(d);
    const dim3 g(nt < c->num_cus ? nt : c->num_cus), b(TL_THREADS);
    if (d.nb <= 8) {
        if (inv)
            SOT_LAUNCH(c, kid, (k_top<true, 8>), g, b, Z, d, c->TW, up, nt);
        else
            SOT_LAUNCH(c, kid, (k_top<false, 8>), g, b, Z, d, c->TW, up, nt);
    } else {
        if (inv)
            SOT_LAUNCH(c, kid, (k_top<true, 9>), g, b, Z, d, c->TW, up, nt);
        else
            SOT_LAUNCH(c, kid, (k_top<false, 9>), g, b, Z, d, c->TW, up, nt);
    }
}

// forward butterflies of an operand (upper half of the row zero) on Z, up to (not including) the low kernel
static inline unsigned mid_grid(const sot_ctx *c, const tile_desc &d) {
    const unsigned np = tile_grid(d) / d.rhalves, cap = 2 * c->num_cus;  // tile pairs; two 64 KiB workgroups per CU
    return np < cap ? np : cap;
}
static void fft_forward_upper(sot_ctx *c, uint32_t *Z, const fft_plan &p, int upto = 3) {
    tile_desc top = p.top;
    top.zero_bit = 0;  // tile bit 0 = level r = 1 = the upper half of the row
    launch_top(c, KID_FFT_TOP, false, top, Z, p.up_fwd);
    if (upto >= 2)
        SOT_LAUNCH(c, KID_FFT_MID, (k_mid<false, false>), dim3(mid_grid(c, p.mid1)), dim3(MD_THREADS), Z, p.mid1, p.mid1_hi, c->MT1, tile_grid(p.mid1), qc_next_counter(c));
    if (upto >= 3) {
        SOT_LAUNCH(c, KID_FFT_MID, (k_mid<false, true>), dim3(mid_grid(c, p.mid2)), dim3(MD_THREADS), Z, p.mid2, p.mid2_hi, c->MT2, tile_grid(p.mid2), qc_next_counter(c));
        for (auto &d : p.w32) launch_w32(c, KID_FFT_MID, false, d, Z, p.up_none);
    }
}

static void fft_inverse_upper(sot_ctx *c, uint32_t *Z, const fft_plan &p) {
    for (size_t i = p.w32.size(); i-- > 0;) launch_w32(c, KID_IFFT_MID, true, p.w32[i], Z, p.up_none);
    SOT_LAUNCH(c, KID_IFFT_MID, (k_mid_inv<true>), dim3(mid_grid(c, p.mid2)), dim3(MD_THREADS), Z, p.mid2, p.mid2_hi, c->MT2, tile_grid(p.mid2), qc_next_counter(c));
    SOT_LAUNCH(c, KID_IFFT_MID, (k_mid_inv<false>), dim3(mid_grid(c, p.mid1)), dim3(MD_THREADS), Z, p.mid1, p.mid1_hi, c->MT1, tile_grid(p.mid1), qc_next_counter(c));
    launch_top(c, KID_IFFT_TOP, true, p.top, Z, p.up_inv);
}

// a_s, s = 1 .. scaler - 1: the first n64 u64 of AesRng(seed = Block::from(s)) (init_a_polynomials,
// quasi_cyclic_encode.rs:127-138), folded mod x^P - 1, its tail bits, and the multiply tables of its transform
static int upload_a_and_transform(sot_ctx *c, uint32_t sidx) {
    const sot_conf &cf = c->conf;
    const qc_geom &g = c->g;
    const uint64_t nb128 = cf.n_blocks * 128;
    uint32_t *const d_a2 = c->a2 + (uint64_t)(sidx - 1) * c->a2_stride;
    uint32_t *const d_ahi = c->a_hi + (uint64_t)(sidx - 1) * c->ahi_stride;
    uint32_t *const d_amat = c->Amat + (uint64_t)(sidx - 1) * c->amat_stride;
    // the raw keystream is staged in Z0 (not yet in use; n_blocks * 16 B is far below its size)
    blk_t *d_araw = (blk_t *)c->Z0;
    if (cf.n_blocks * 16 > g.nw * ROWS * 4) {
        sot_set_error("internal: a does not fit the staging buffer");
        return SOT_EINVAL;
    }
    uint8_t key[16] = {(uint8_t)sidx, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    SOT_TRY(util_aes_ctr_device(c, key, 0, d_araw, cf.n_blocks));
    std::vector<uint32_t> a(cf.n_blocks * 4 + 2);
    SOT_HIP(hipMemcpy(a.data(), d_araw, cf.n_blocks * 16, hipMemcpyDeviceToHost));
    // a' = a mod (x^P - 1): fold bits [P, nb128) onto [0, nb128 - P)
    auto getbit = [&](uint64_t i) -> uint32_t { return (a[i >> 5] >> (i & 31)) & 1u; };
    const uint64_t P = cf.P;
    std::vector<uint32_t> ap((P + 31) / 32 + 2, 0);
    for (uint64_t w = 0; w < (P + 31) / 32; w++) ap[w] = a[w];
    if (P & 31) ap[P / 32] &= (1u << (P & 31)) - 1u;
    for (uint64_t i = P; i < nb128; i++)
        if (getbit(i)) ap[(i - P) >> 5] ^= 1u << ((i - P) & 31);
    // tail bits a'[S .. P)
    std::vector<uint32_t> ahi((g.tail + 31) / 32 + 1, 0);
    for (uint64_t i = 0; i < g.tail; i++)
        if ((ap[(g.S + i) >> 5] >> ((g.S + i) & 31)) & 1u) ahi[i >> 5] |= 1u << (i & 31);
    SOT_HIP(hipMemcpy(d_a2, ap.data(), ap.size() * 4, hipMemcpyHostToDevice));
    SOT_HIP(hipMemcpy(d_ahi, ahi.data(), ahi.size() * 4, hipMemcpyHostToDevice));
    // FFT(a_lo): row 0 of Z
    const fft_plan &plan = *(const fft_plan *)c->qc_plan;
    SOT_HIP(hipMemsetAsync(c->Z0, 0, g.nw * ROWS * 4, c->stream));
    SOT_HIP(hipMemsetAsync(c->Z1, 0, g.nw * ROWS * 4, c->stream));
    hipLaunchKernelGGL(k_a_load, dim3(gridn(g.nw / 2, 256)), dim3(256), 0, c->stream, d_a2, g.lo_bits, g.nw / 2, c->Z0);
    uint32_t *zb[2] = {c->Z0, c->Z1};
    int cur = run_conversion(c, zb, 0, g.m - 1, false, plan.units_in_top, &plan.sub_fwd, &plan.sub_inv);
    fft_forward_upper(c, zb[cur], plan);
    if (sidx == 1)
        hipLaunchKernelGGL(k_build_lowmasks, dim3(gridn((g.nw / 32) * 5 * LOW_BLK_PLAIN, 256)), dim3(256), 0, c->stream, plan.low,
                           g.nw / 32, c->TW, c->LM);
    launch_low<0>(c, zb[cur], plan.low);
    hipLaunchKernelGGL(k_build_amat, dim3(gridn((g.nw / 32) * AMAT_PLAIN, 256)), dim3(256), 0, c->stream, zb[cur], plan.low,
                       g.nw / 32, d_amat);
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
    // FFT capacity: 2^m product bits, S = 2^(m-1) bits per operand; up to TAIL_MAX operand bits above S go
    // through the tail terms of k_qc_finish. m >= 21 keeps all five pruned levels on word-index bits.
    const uint64_t TAIL_MAX = 256;
    uint32_t m = 21;
    while ((1ull << (m - 1)) + TAIL_MAX < cf.P) m++;
    if (m > 30) {
        sot_set_error("n too large for the GF(2^32) transform");
        return SOT_EINVAL;
    }
    g.m = m;
    g.S = 1ull << (m - 1);
    g.tail = cf.P > g.S ? cf.P - g.S : 0;
    g.lo_bits = cf.P < g.S ? cf.P : g.S;
    g.nw = 1ull << (m - 5);
    SOT_TRY(dev_alloc(c, (void **)&c->Z0, g.nw * ROWS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->Z1, g.nw * ROWS * 4));
    const uint64_t nseg = cf.scaler - 1;  // public polynomials a_1 .. a_{scaler-1}
    c->amat_stride = (g.nw / 32) * AMAT_WORDS, c->a2_stride = ((cf.P + 31) / 32 + 2 + 3) & ~3ull,
    c->ahi_stride = (g.tail + 31) / 32 + 1;
    SOT_TRY(dev_alloc(c, (void **)&c->Amat, nseg * c->amat_stride * 4));
    c->Amat_cur = c->Amat;
    SOT_TRY(dev_alloc(c, (void **)&c->LM, (g.nw / 32) * 5 * LOW_BLK_WORDS * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->a2, nseg * c->a2_stride * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->sbv, (cf.cols * 128 / 32 + 2) * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->a_hi, nseg * c->ahi_stride * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->TW, TW_TABLES * 256 * 32 * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->TWC, TW_TABLES * 256 * 4));
    hipLaunchKernelGGL(k_build_tw, dim3(TW_TABLES * 256 * 32 / 256), dim3(256), 0, c->stream, c->TW, c->TWC);
    SOT_HIP(hipGetLastError());
    fft_plan *plan = new fft_plan(make_plan(m, c->opts));
    c->qc_plan = plan;
    SOT_TRY(dev_alloc(c, (void **)&c->d_ctr, QC_NCTR * sizeof(uint32_t)));
    SOT_HIP(hipMemsetAsync(c->d_ctr, 0, QC_NCTR * sizeof(uint32_t), c->stream));
    c->ctr_next = 0;
    SOT_TRY(dev_alloc(c, (void **)&c->MT1, 16ull * 4 * 8 * 256 * 4));
    SOT_TRY(dev_alloc(c, (void **)&c->MT2, 256ull * 4 * 8 * 256 * 4));
    hipLaunchKernelGGL(k_build_midmasks, dim3(16 * 4 * 8 * 16 / 256), dim3(256), 0, c->stream, plan->mid1, plan->mid1_hi, 9u,
                       16u, c->TW, c->MT1);
    hipLaunchKernelGGL(k_build_midmasks, dim3(256 * 4 * 8 * 16 / 256), dim3(256), 0, c->stream, plan->mid2, plan->mid2_hi,
                       13u, 256u, c->TW, c->MT2);
    SOT_HIP(hipGetLastError());
    {
        const size_t nf = plan->list_fwd.size(), ni = plan->list_inv.size();
        const size_t sf = plan->sub_list_fwd.size(), si = plan->sub_list_inv.size();
        SOT_TRY(dev_alloc(c, (void **)&c->up_list, (nf + ni + sf + si + 1) * 4));
        if (nf) SOT_HIP(hipMemcpy(c->up_list, plan->list_fwd.data(), nf * 4, hipMemcpyHostToDevice));
        if (ni) SOT_HIP(hipMemcpy(c->up_list + nf, plan->list_inv.data(), ni * 4, hipMemcpyHostToDevice));
        SOT_HIP(hipMemcpy(c->up_list + nf + ni, plan->sub_list_fwd.data(), sf * 4, hipMemcpyHostToDevice));
        SOT_HIP(hipMemcpy(c->up_list + nf + ni + sf, plan->sub_list_inv.data(), si * 4, hipMemcpyHostToDevice));
        plan->up_fwd.list = c->up_list;
        plan->up_inv.list = c->up_list + nf;
        plan->sub_fwd.list = c->up_list + nf + ni;
        plan->sub_inv.list = c->up_list + nf + ni + sf;
    }
    for (uint32_t sidx = 1; sidx < cf.scaler; sidx++) SOT_TRY(upload_a_and_transform(c, sidx));
    return SOT_OK;
}

// A sharded context of 4 or more ranks holds at most 64 bit rows (2 x 128 / world, both roles): the QC kernels then skip
// the upper row half of every gather tile and the 16-row slices beyond `rows` in the conversion passes; k_qc_finish reads
// the skipped rows as zero (they ARE zero: the encode is linear and those input rows are).
void qc_set_rows_active(sot_ctx *c, uint32_t rows) {
    c->rows_active = rows < ROWS ? rows : ROWS;
    fft_plan *p = (fft_plan *)c->qc_plan;
    const uint32_t rh = c->rows_active > TL_ROWS ? 2 : 1;
    p->top.rhalves = p->mid1.rhalves = p->mid2.rhalves = p->low.rhalves = rh;
    for (auto &d : p->w32) d.rhalves = rh;
}

void qc_free(sot_ctx *c) {
    delete (fft_plan *)c->qc_plan;
    c->qc_plan = nullptr;
    void *ptrs[] = {c->Z0, c->Z1, c->Amat, c->sbv, c->a2, c->a_hi, c->TW, c->TWC, c->up_list, c->LM, c->MT1, c->MT2, c->d_ctr};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
}

// noise indicator bit vector sb (one bit per leaf Block) from the receiver's interleaved points
static int qc_set_points(sot_ctx *c, const uint64_t *points, uint64_t npoints) {
    const uint64_t nV = c->conf.cols * 128;
    SOT_TRY(stage_h2d(c, c->d_points, points, npoints * 8));
    SOT_HIP(hipMemsetAsync(c->sbv, 0, (nV / 32 + 2) * 4, c->stream));
    SOT_LAUNCH(c, KID_CLEAR_LSB, k_set_points, dim3((npoints + 127) / 128), dim3(128), c->sbv, c->d_points,
               (uint32_t)npoints, nV);
    return SOT_OK;
}

// leaves V (Blocks, or slot Blocks of a sharded context) -> encoded Blocks out[0 .. n): load, forward transform,
// pointwise multiply, inverse transform, finish. Asynchronous on the context's stream.
static int qc_encode_core(sot_ctx *c, const blk_t *V, const uint32_t *sb, uint32_t sbpos, blk_t *out, uint32_t out_words = 4) {
    const sot_conf &cf = c->conf;
    const qc_geom &g = c->g;
    const uint64_t nb128 = cf.n_blocks * 128;
    const fft_plan &plan = *(const fft_plan *)c->qc_plan;
    if (g.tail > FN_TAILMAX) {
        sot_set_error("internal: tail exceeds FN_TAILMAX");
        return SOT_EINVAL;
    }
    const bool full = c->rows_active == ROWS && out_words == 4;
    // c = b0 + sum_{s = 1}^{scaler - 1} a_s b_s (MultAddReducer::reduce, quasi_cyclic_encode.rs:262-286), one product per pass:
    // the pass for s sees the leaves re-based by (s - 1) nb128 Blocks, so that b_s is its "V1"; the first pass adds the b0
    // terms and writes `out`, the later ones (scaler > 2) XOR their product into it -- the transform is linear, so summing
    // after the inverse transforms equals the reference's sum in the transform domain.
    for (uint32_t sidx = 1; sidx < cf.scaler; sidx++) {
        const blk_t *Vs = (const blk_t *)((const uint32_t *)V + (uint64_t)(sidx - 1) * nb128 * out_words);
        const uint32_t *sbs = sb ? sb + (uint64_t)(sidx - 1) * nb128 / 32 : nullptr;
        const blk_t *V0 = sidx == 1 ? V : nullptr;
        c->Amat_cur = c->Amat + (uint64_t)(sidx - 1) * c->amat_stride;
        if (full)
            SOT_LAUNCH(c, KID_QC_LOAD, k_qc_load<true>, dim3(gridn(g.nw / 2, LD_W)), dim3(LD_T), Vs, sbs, sbpos, nb128, cf.P, g.lo_bits,
                       g.nw / 2, c->Z0, c->rows_active, out_words);
        else
            SOT_LAUNCH(c, KID_QC_LOAD, k_qc_load<false>, dim3(gridn(g.nw / 2, LD_W)), dim3(LD_T), Vs, sbs, sbpos, nb128, cf.P, g.lo_bits,
                       g.nw / 2, c->Z0, c->rows_active, out_words);
        uint32_t *zb[2] = {c->Z0, c->Z1};
        int cur = run_conversion(c, zb, 0, g.m - 1, false, plan.units_in_top, &plan.sub_fwd, &plan.sub_inv);
        fft_forward_upper(c, zb[cur], plan);
        if (c->prof) prof_begin(c, KID_FFT_LOW_MUL);
        launch_low<1>(c, zb[cur], plan.low);
        if (c->prof) prof_end(c, KID_FFT_LOW_MUL);
        fft_inverse_upper(c, zb[cur], plan);
        cur = run_conversion(c, zb, cur, g.m, true, plan.units_in_top, &plan.sub_fwd, &plan.sub_inv);
        const uint32_t *a2s = c->a2 + (uint64_t)(sidx - 1) * c->a2_stride, *ahis = c->a_hi + (uint64_t)(sidx - 1) * c->ahi_stride;
        if (c->prof) prof_begin(c, KID_QC_FINAL);
        if (full)
            hipLaunchKernelGGL(k_qc_finish<true>, dim3(gridn(cf.n, FN_K)), dim3(256), fn_dyn_lds((uint32_t)g.tail), c->stream, zb[cur],
                               g.nw, Vs, sbs, sbpos, nb128, cf.P, g.S, (uint32_t)g.tail, cf.n, a2s, ahis, out, c->rows_active,
                               out_words, V0, sb);
        else
            hipLaunchKernelGGL(k_qc_finish<false>, dim3(gridn(cf.n, FN_K)), dim3(256), fn_dyn_lds((uint32_t)g.tail), c->stream, zb[cur],
                               g.nw, Vs, sbs, sbpos, nb128, cf.P, g.S, (uint32_t)g.tail, cf.n, a2s, ahis, out, c->rows_active,
                               out_words, V0, sb);
        if (c->prof) prof_end(c, KID_QC_FINAL);
    }
    c->Amat_cur = c->Amat;
    SOT_HIP(hipGetLastError());
    return SOT_OK;
}

int qc_encode(sot_ctx *c, int role, const uint64_t *points, uint64_t npoints, int packed, uint8_t *C_out) {
    const sot_conf &cf = c->conf;
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nb128 = cf.n_blocks * 128;
    const uint32_t *sb = nullptr;
    if (role == SOT_ROLE_RECEIVER && !packed) {
        // the choice vector: sparse sum of shifts of a' (k_sparse_choice); A itself is the plain encode of all 128 rows
        if (cf.n > c->tmp_bytes) {  // cannot happen: the staging buffer holds n x 32 bytes
            sot_set_error("internal: the staging buffer is smaller than the choice vector (%llu bytes)", (unsigned long long)cf.n);
            return SOT_EINVAL;
        }
        std::vector<uint64_t> e(npoints);
        for (uint64_t i = 0; i < npoints; i++) {
            const uint64_t S = points[i], seg = S / nb128;
            if (seg == 0)
                e[i] = (S % cf.P) | (1ull << 63);
            else if (seg < cf.scaler)
                e[i] = ((S - seg * nb128) % cf.P) | ((seg - 1) << 56);
            else
                e[i] = SC_DROPPED;
        }
        SOT_HIP(hipMemcpyAsync(c->d_points, e.data(), npoints * 8, hipMemcpyHostToDevice, c->stream));
        SOT_LAUNCH(c, KID_CLEAR_LSB, k_sparse_choice, dim3(gridn((cf.n + 31) / 32, 256)), dim3(256), c->a2, c->a2_stride, cf.P, c->d_points,
                   (uint32_t)npoints, cf.n, (uint8_t *)c->d_tmp);
        SOT_HIP(hipGetLastError());
        SOT_HIP(hipMemcpyAsync(C_out, c->d_tmp, cf.n, hipMemcpyDeviceToHost, c->stream));
        SOT_HIP(hipStreamSynchronize(c->stream));  // e goes out of scope; C_out is complete
    } else if (role == SOT_ROLE_RECEIVER) {
        SOT_TRY(qc_set_points(c, points, npoints));
        sb = c->sbv;
    }
    SOT_TRY(qc_encode_core(c, c->role[role].V, sb, 0, c->role[role].enc));
    SOT_SYNC(c);
    c->role[role].encoded = true;
    if (role == SOT_ROLE_RECEIVER) c->recv_unpacked = !packed;
    return SOT_OK;
}

// Sharded context: ONE pass over the slot Blocks W encodes this rank's bit rows of the sender AND the receiver matrix
// (the 128 rows of a wave's two halves are 2 x 128 / world sender + receiver rows; both multiply the same public a, so
// every mask is shared). The receiver's bit row 0 -- the row dual_encode2 replaces by sb -- is slot 128 / world of rank 0.
int qc_encode_slots(sot_ctx *c, const uint64_t *points, uint64_t npoints, blk_t *out) {
    SOT_HIP(hipSetDevice(c->device));
    const uint32_t *sb = nullptr;
    if (c->sh.rank == 0) {
        SOT_TRY(qc_set_points(c, points, npoints));
        sb = c->sbv;
    }
    // 2 x rpr live bits per output: whole Blocks at 2 ranks, 8 bytes at 4, 4 bytes at 8
    SOT_TRY(qc_encode_core(c, c->sh.W, sb, c->sh.rpr, out, c->sh.rpr >= 64 ? 4 : c->sh.rpr >= 32 ? 2 : 1));
    SOT_SYNC(c);
    return SOT_OK;
}

// ---- debug / validation hook: runs the stages of the transform on ONE row supplied by the host (row 0 of Z; the
// other rows are zero) so that tests can compare every stage with fafft_model.py.
//   stage 0: forward basis conversion of the low half     stage 1: + top pass (levels 1..8)
//   stage 2: + levels 9..12   stage 3: + levels 13..16 (and the extra 32-plane passes)   stage 4: + low levels (full
//   forward transform)
int qc_debug_row(sot_ctx *c, const uint32_t *row_in, int stage, uint32_t *row_out) {
    const qc_geom &g = c->g;
    SOT_HIP(hipSetDevice(c->device));
    const fft_plan &plan = *(const fft_plan *)c->qc_plan;
    SOT_HIP(hipMemsetAsync(c->Z0, 0, g.nw * ROWS * 4, c->stream));
    SOT_HIP(hipMemsetAsync(c->Z1, 0, g.nw * ROWS * 4, c->stream));
    SOT_HIP(hipMemcpy2DAsync(c->Z0, ROWS * 4, row_in, 4, 4, g.nw / 2, hipMemcpyHostToDevice, c->stream));
    uint32_t *zb[2] = {c->Z0, c->Z1};
    int cur = run_conversion(c, zb, 0, g.m - 1, false, stage >= 1 && plan.units_in_top, &plan.sub_fwd, &plan.sub_inv);
    uint32_t *Z = zb[cur];
    if (stage >= 1) fft_forward_upper(c, Z, plan, stage);
    if (stage >= 4)
        launch_low<0>(c, Z, plan.low);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipMemcpy2DAsync(row_out, 4, Z, ROWS * 4, 4, g.nw, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    return SOT_OK;
}
