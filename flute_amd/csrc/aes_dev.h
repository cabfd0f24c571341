// Device-side AES-128 (FIPS-197) with the four round T-tables staged in LDS.
// Reference behaviour: crate aes 0.8.3 as used at zappot/src/silent_ot/pprf.rs:175-180,
// util/aes_hash.rs:47-55 and util/aes_rng.rs:121.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint4 blk_t;

static __device__ const uint32_t d_TE0[256] = {
#include "aes_te0.inc"
};
// expanded fixed keys: [0] = GGM left child k0, [1] = GGM right child k1 (pprf.rs:675-688),
// [2] = correlation-robust hash key (aes_hash.rs:59-66)
static __constant__ uint32_t c_rk_fixed[3][44] = {
#include "aes_rk_fixed.inc"
};
#define AES_K0 0
#define AES_K1 1
#define AES_KH 2

// LDS image: the single table TE0 replicated 32x, entry e of replica r at word e*32 + r. A lane reads replica
// (lane & 31), so every ds_read_b32 of a wave hits 32 distinct banks in each 32-lane half: no bank conflicts,
// whatever the (secret) byte values are. TE1..TE3 are rotations of TE0 (v_alignbit).
#define AES_LDS_WORDS (256 * 32)
typedef uint32_t aes_lds_t[1][AES_LDS_WORDS];

__device__ __forceinline__ void aes_lds_fill(uint32_t (*te)[AES_LDS_WORDS]) {
    for (uint32_t i = threadIdx.x; i < AES_LDS_WORDS; i += blockDim.x) te[0][i] = d_TE0[i >> 5];
    __syncthreads();
}

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ blk_t aes_encrypt(const uint32_t (*te)[AES_LDS_WORDS], const uint32_t *__restrict__ rk,
                                             blk_t in) {
    const uint32_t *T = te[0] + (threadIdx.x & 31);
#define TE(b) T[(b) << 5]
    uint32_t s0 = in.x ^ rk[0], s1 = in.y ^ rk[1], s2 = in.z ^ rk[2], s3 = in.w ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        uint32_t t0 = TE(s0 & 255) ^ rotl32(TE((s1 >> 8) & 255), 8) ^ rotl32(TE((s2 >> 16) & 255), 16) ^
                      rotl32(TE(s3 >> 24), 24) ^ rk[4 * r];
        uint32_t t1 = TE(s1 & 255) ^ rotl32(TE((s2 >> 8) & 255), 8) ^ rotl32(TE((s3 >> 16) & 255), 16) ^
                      rotl32(TE(s0 >> 24), 24) ^ rk[4 * r + 1];
        uint32_t t2 = TE(s2 & 255) ^ rotl32(TE((s3 >> 8) & 255), 8) ^ rotl32(TE((s0 >> 16) & 255), 16) ^
                      rotl32(TE(s1 >> 24), 24) ^ rk[4 * r + 2];
        uint32_t t3 = TE(s3 & 255) ^ rotl32(TE((s0 >> 8) & 255), 8) ^ rotl32(TE((s1 >> 16) & 255), 16) ^
                      rotl32(TE(s2 >> 24), 24) ^ rk[4 * r + 3];
        s0 = t0;
        s1 = t1;
        s2 = t2;
        s3 = t3;
    }
    // last round: SubBytes + ShiftRows only; TE0 bytes are (2S, S, S, 3S): byte 1 (and 2) hold S[x]
    blk_t o;
    o.x = (((TE(s0 & 255) >> 8) & 0xffu) | (TE((s1 >> 8) & 255) & 0xff00u) | (TE((s2 >> 16) & 255) & 0xff0000u) |
           ((TE(s3 >> 24) << 8) & 0xff000000u)) ^
          rk[40];
    o.y = (((TE(s1 & 255) >> 8) & 0xffu) | (TE((s2 >> 8) & 255) & 0xff00u) | (TE((s3 >> 16) & 255) & 0xff0000u) |
           ((TE(s0 >> 24) << 8) & 0xff000000u)) ^
          rk[41];
    o.z = (((TE(s2 & 255) >> 8) & 0xffu) | (TE((s3 >> 8) & 255) & 0xff00u) | (TE((s0 >> 16) & 255) & 0xff0000u) |
           ((TE(s1 >> 24) << 8) & 0xff000000u)) ^
          rk[42];
    o.w = (((TE(s3 & 255) >> 8) & 0xffu) | (TE((s0 >> 8) & 255) & 0xff00u) | (TE((s1 >> 16) & 255) & 0xff0000u) |
           ((TE(s2 >> 24) << 8) & 0xff000000u)) ^
          rk[43];
#undef TE
    return o;
}

__device__ __forceinline__ blk_t bxor(blk_t a, blk_t b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ blk_t bnot(blk_t a) { return make_uint4(~a.x, ~a.y, ~a.z, ~a.w); }
__device__ __forceinline__ blk_t bzero() { return make_uint4(0, 0, 0, 0); }

// H(x) = AES_kh(x) ^ x  (util/aes_hash.rs:26-30)
__device__ __forceinline__ blk_t cr_hash(const uint32_t (*te)[AES_LDS_WORDS], blk_t x) {
    return bxor(aes_encrypt(te, c_rk_fixed[AES_KH], x), x);
}
