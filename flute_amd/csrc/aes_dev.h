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

// te[i][x] = rotl(TE0[x], 8 i)
__device__ __forceinline__ void aes_lds_fill(uint32_t (*te)[256]) {
    for (uint32_t i = threadIdx.x; i < 256; i += blockDim.x) {
        uint32_t t = d_TE0[i];
        te[0][i] = t;
        te[1][i] = (t << 8) | (t >> 24);
        te[2][i] = (t << 16) | (t >> 16);
        te[3][i] = (t << 24) | (t >> 8);
    }
    __syncthreads();
}

__device__ __forceinline__ blk_t aes_encrypt(const uint32_t (*te)[256], const uint32_t *__restrict__ rk, blk_t in) {
    uint32_t s0 = in.x ^ rk[0], s1 = in.y ^ rk[1], s2 = in.z ^ rk[2], s3 = in.w ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
        uint32_t t0 = te[0][s0 & 255] ^ te[1][(s1 >> 8) & 255] ^ te[2][(s2 >> 16) & 255] ^ te[3][s3 >> 24] ^ rk[4 * r];
        uint32_t t1 =
            te[0][s1 & 255] ^ te[1][(s2 >> 8) & 255] ^ te[2][(s3 >> 16) & 255] ^ te[3][s0 >> 24] ^ rk[4 * r + 1];
        uint32_t t2 =
            te[0][s2 & 255] ^ te[1][(s3 >> 8) & 255] ^ te[2][(s0 >> 16) & 255] ^ te[3][s1 >> 24] ^ rk[4 * r + 2];
        uint32_t t3 =
            te[0][s3 & 255] ^ te[1][(s0 >> 8) & 255] ^ te[2][(s1 >> 16) & 255] ^ te[3][s2 >> 24] ^ rk[4 * r + 3];
        s0 = t0;
        s1 = t1;
        s2 = t2;
        s3 = t3;
    }
    // last round: SubBytes + ShiftRows only. te2 byte0 = te3 byte1 = te0 byte2 = te1 byte3 = S[x]
    blk_t o;
    o.x = ((te[2][s0 & 255] & 0xffu) | (te[3][(s1 >> 8) & 255] & 0xff00u) | (te[0][(s2 >> 16) & 255] & 0xff0000u) |
           (te[1][s3 >> 24] & 0xff000000u)) ^
          rk[40];
    o.y = ((te[2][s1 & 255] & 0xffu) | (te[3][(s2 >> 8) & 255] & 0xff00u) | (te[0][(s3 >> 16) & 255] & 0xff0000u) |
           (te[1][s0 >> 24] & 0xff000000u)) ^
          rk[41];
    o.z = ((te[2][s2 & 255] & 0xffu) | (te[3][(s3 >> 8) & 255] & 0xff00u) | (te[0][(s0 >> 16) & 255] & 0xff0000u) |
           (te[1][s1 >> 24] & 0xff000000u)) ^
          rk[42];
    o.w = ((te[2][s3 & 255] & 0xffu) | (te[3][(s0 >> 8) & 255] & 0xff00u) | (te[0][(s1 >> 16) & 255] & 0xff0000u) |
           (te[1][s2 >> 24] & 0xff000000u)) ^
          rk[43];
    return o;
}

__device__ __forceinline__ blk_t bxor(blk_t a, blk_t b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ blk_t bnot(blk_t a) { return make_uint4(~a.x, ~a.y, ~a.z, ~a.w); }
__device__ __forceinline__ blk_t bzero() { return make_uint4(0, 0, 0, 0); }

// H(x) = AES_kh(x) ^ x  (util/aes_hash.rs:26-30)
__device__ __forceinline__ blk_t cr_hash(const uint32_t (*te)[256], blk_t x) {
    return bxor(aes_encrypt(te, c_rk_fixed[AES_KH], x), x);
}
