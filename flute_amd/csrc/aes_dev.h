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
// [2] = correlation-robust hash key (aes_hash.rs:59-66). 88 words per key: the 44 round-key words, then the
// same words rotated right by 8 bits (consumed by the main rounds, see aes_encrypt).
#define AES_RK_WORDS 88
static __constant__ uint32_t c_rk_fixed[3][AES_RK_WORDS] = {
#include "aes_rk_fixed.inc"
};
#define AES_K0 0
#define AES_K1 1
#define AES_KH 2

// LDS image (64 KiB): one 256-byte row per table index e; words 0..31 of the row are 32 replicas of TE0[e], words
// 32..63 are 32 replicas of TE2[e] = TE0[e] rotated left by 16. A lane reads replica (lane & 31), so every
// ds_read_b32 of a wave hits 32 distinct banks in each 32-lane half: no bank conflicts, whatever the (secret) byte
// values are. The byte address of a lookup, (e << 8) | ((lane & 31) << 2), is ONE v_perm_b32 of the state word and
// the lane constant; TE2 is the same address + 128 (the instruction's immediate offset).
#define AES_LDS_WORDS (256 * 64)
typedef uint32_t aes_lds_t[1][AES_LDS_WORDS];

__device__ __forceinline__ uint32_t rotl32(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, 32 - n); }

__device__ __forceinline__ void aes_lds_fill(uint32_t (*te)[AES_LDS_WORDS]) {
    // four replicas per 16-byte store (words 4 i .. 4 i + 3 lie in the same half row)
    uint4 *t4 = (uint4 *)te[0];
    for (uint32_t i = threadIdx.x; i < AES_LDS_WORDS / 4; i += blockDim.x) {
        const uint32_t t = d_TE0[i >> 4];
        const uint32_t v = (i & 8) ? rotl32(t, 16) : t;
        t4[i] = make_uint4(v, v, v, v);
    }
    __syncthreads();
}

#define XOR3(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x96)
#define OR_XOR(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x56)  // (a | b) ^ c

// rk: AES_RK_WORDS words (plain, then rotated right by 8); ROT_INLINE: only the 44 plain words exist (per-lane keys
// kept in LDS, alsz.hip) and the rotation is one v_alignbit per column
template <bool ROT_INLINE = false>
__device__ __forceinline__ blk_t aes_encrypt(const uint32_t (*te)[AES_LDS_WORDS], const uint32_t *__restrict__ rk,
                                             blk_t in) {
    const char *tb = (const char *)te[0];
    const uint32_t loff = (threadIdx.x & 31) << 2;
    // v_perm_b32: selector 0-3 = bytes of the second operand, 4-7 = bytes of the first, 0x0c = zero
#define ADR(s, k) __builtin_amdgcn_perm((s), loff, 0x0c0c0000u | ((4u + (k)) << 8))
#define T0(s, k) (*(const uint32_t *)(tb + ADR(s, k)))
#define T2(s, k) (*(const uint32_t *)(tb + ADR(s, k) + 128))
    // one column of a full round: TE0[a] ^ rotl8(TE0[b]) ^ rotl16(TE0[c]) ^ rotl24(TE0[d]) ^ k
    //                           = TE0[a] ^ TE2[c] ^ rotl8(TE0[b] ^ TE2[d] ^ rotr8(k))
#define COL(a, b, c, d, kr) XOR3(T0(a, 0), T2(c, 2), rotl32(XOR3(T0(b, 1), T2(d, 3), (kr)), 8))
    uint32_t s0 = in.x ^ rk[0], s1 = in.y ^ rk[1], s2 = in.z ^ rk[2], s3 = in.w ^ rk[3];
#pragma unroll
    for (int r = 1; r < 10; r++) {
#define RKR(i) (ROT_INLINE ? rotl32(rk[(i)], 24) : rk[44 + (i)])
        const uint32_t t0 = COL(s0, s1, s2, s3, RKR(4 * r));
        const uint32_t t1 = COL(s1, s2, s3, s0, RKR(4 * r + 1));
        const uint32_t t2 = COL(s2, s3, s0, s1, RKR(4 * r + 2));
        const uint32_t t3 = COL(s3, s0, s1, s2, RKR(4 * r + 3));
#undef RKR
        s0 = t0, s1 = t1, s2 = t2, s3 = t3;
    }
    // last round: SubBytes + ShiftRows only. TE0 bytes are (2S, S, S, 3S): bytes 1 and 2 hold S[x]; two byte
    // permutes gather them
#define LAST(a, b, c, d, k)                                                          \
    OR_XOR(__builtin_amdgcn_perm(T0(b, 1), T0(a, 0), 0x0c0c0501u),                   \
           __builtin_amdgcn_perm(T0(d, 3), T0(c, 2), 0x06020c0cu), (k))
    blk_t o;
    o.x = LAST(s0, s1, s2, s3, rk[40]);
    o.y = LAST(s1, s2, s3, s0, rk[41]);
    o.z = LAST(s2, s3, s0, s1, rk[42]);
    o.w = LAST(s3, s0, s1, s2, rk[43]);
#undef LAST
#undef COL
#undef T2
#undef T0
#undef ADR
    return o;
}

__device__ __forceinline__ blk_t bxor(blk_t a, blk_t b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
__device__ __forceinline__ blk_t bnot(blk_t a) { return make_uint4(~a.x, ~a.y, ~a.z, ~a.w); }
__device__ __forceinline__ blk_t bzero() { return make_uint4(0, 0, 0, 0); }

// H(x) = AES_kh(x) ^ x  (util/aes_hash.rs:26-30)
__device__ __forceinline__ blk_t cr_hash(const uint32_t (*te)[AES_LDS_WORDS], blk_t x) {
    return bxor(aes_encrypt(te, c_rk_fixed[AES_KH], x), x);
}
