// 128 x 128 bit-matrix transpose held across the 128 threads of two consecutive wavefronts (thread r owns row r as four
// 32-bit words), LSB first: on return thread r holds column r, i.e. w[g] bit b = (old word r >> 5, bit r & 31 of thread
// 32 g + b). util/transpose.rs:53-109 semantics (out row r bit j = in row j bit r).
//
// Two steps instead of 128 single-bit LDS reads per thread (round 2):
//   1. the sixteen 32 x 32 blocks (32 consecutive lanes x one word) are transposed in place by the five-stage exchange
//      network, one cross-lane move + one rotate + one bit-select per word and stage: partners at lane distance 1, 2, 8 by
//      DPP, 4 and 16 by ds_swizzle (no LDS memory access);
//   2. word k of lane group g now belongs to thread 32 k + (lane & 31) as its word g: a 4 x 4 exchange of words through
//      512 words of LDS (4 conflict-free ds_write_b32 + 4 ds_read_b32 per thread, one workgroup barrier).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int J>
__device__ __forceinline__ uint32_t bm_xlane(uint32_t v) {  // the value of lane ^ J (J < 32: stays inside a 32-lane group)
    if (J == 1) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    if (J == 2) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    if (J == 8) return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x128, 0xF, 0xF, true);   // row_ror:8
    return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, (J << 10) | 0x1F);                 // bit mode: lane ^ J
}

template <int J, int N = 4>
__device__ __forceinline__ void bm_stage(uint32_t *w, uint32_t lane) {
    constexpr uint32_t M = J == 16 ? 0x0000FFFFu : J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
    const bool hi = (lane & J) != 0;
    const uint32_t keep = hi ? ~M : M;     // own bits that stay
    const uint32_t rot = hi ? J : 32 - J;  // partner bits: down by J (upper lane) or up by J (lower lane)
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t p = bm_xlane<J>(w[k]);
        const uint32_t r = __builtin_amdgcn_alignbit(p, p, rot);
        w[k] = __builtin_amdgcn_bitop3_b32(keep, w[k], r, 0xCA);  // keep ? own : partner
    }
}

// xch: 512 words of LDS private to these 128 threads; r = the thread's row (0 .. 127, lane = r & 63). Contains one
// __syncthreads(): every thread of the workgroup must call it. The caller separates two uses of xch by a barrier.
__device__ __forceinline__ void bm_transpose128(uint32_t w[4], uint32_t *xch, uint32_t r) {
    const uint32_t lane = r & 31u;
    bm_stage<16>(w, lane), bm_stage<8>(w, lane), bm_stage<4>(w, lane), bm_stage<2>(w, lane), bm_stage<1>(w, lane);
    const uint32_t g = r >> 5;
#pragma unroll
    for (int k = 0; k < 4; k++) xch[(k * 4 + g) * 32 + lane] = w[k];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) w[q] = xch[(g * 4 + q) * 32 + lane];
}

// The same tile held by ONE wavefront: lane l owns row l (w[0..3]) and row l + 64 (w[4..7]); on return it holds columns
// l and l + 64 the same way. No LDS memory and no barrier: after the block transposes, word k of row group
// g = 2 s + h (s = register set, h = lane half) belongs to row group k as its word g -- the exchange s <-> k bit 1 is a
// renaming of registers, h <-> k bit 0 swaps the upper lane half of word 2 k1 with the lower lane half of word 2 k1 + 1
// (v_permlane32_swap_b32, gfx950).
__device__ __forceinline__ void bm_transpose128_wave(uint32_t w[8], uint32_t lane64) {
    const uint32_t lane = lane64 & 31u;
    bm_stage<16, 8>(w, lane), bm_stage<8, 8>(w, lane), bm_stage<4, 8>(w, lane), bm_stage<2, 8>(w, lane), bm_stage<1, 8>(w, lane);
    uint32_t o[8];
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int k1 = 0; k1 < 2; k1++) {
            const auto r = __builtin_amdgcn_permlane32_swap(w[4 * s + 2 * k1], w[4 * s + 2 * k1 + 1], false, false);
            o[4 * k1 + 2 * s + 0] = r[0];  // word from lane half 0 of the source group pair
            o[4 * k1 + 2 * s + 1] = r[1];
        }
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = o[k];
}
