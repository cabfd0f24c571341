// GGM / PPRF expansion, tile transposes, CR hashes and MT-bit derivation for gfx950.
//
// Data layout in HBM
//   interior GGM level l : [set][node 0..2^l)[tree%8] Blocks   (set = tree/8, the reference's [Block;8] nodes,
//                          zappot/src/silent_ot/pprf.rs:126)
//   leaves "V"           : V[128*(set + nsets*(leaf/16)) + 8*(leaf%16) + tree%8]  -- exactly the order in which
//                          copy_out (pprf.rs:810-835) feeds 128-Block tiles to the bit transpose, so
//                          rT[r][col] bit j == V[128*col + j] bit r. The QC stage consumes V directly.
//   sums                 : [level][set][keep][tree%8] Blocks
#include "aes_dev.h"
#include "bitmat_dev.h"
#include "sot_internal.h"
#include <string.h>

#ifndef GGM_BLOCK
#define GGM_BLOCK 512  // 8 waves; the 64 KiB AES table allows two workgroups per CU
#endif
#ifndef GGM_EXP
#define GGM_EXP 0  // profiling experiments only (tools/exp_one.sh; results are wrong when set): 1 = levels without their stores
#endif
#define HASH_BLOCK 512

// ------------------------------------------------------------------------------------------------
// roots: worker t = set % T owns AesRng(seed ^ t); its i-th group (i = set / T) takes keystream blocks
// 8i..8i+7 (pprf.rs:115,128,135-137; util/aes_rng.rs:109-122)
// Every per-tree kernel takes the tree range [tr0, tr1) it works on: all trees on one GPU, the trees of this rank's tree
// sets when one expansion is sharded over several GPUs (pprf.rs:128: the reference strides the same groups over workers).
__global__ void k_roots(const uint32_t *__restrict__ rks, uint32_t T, uint32_t tr0, uint32_t tr1, blk_t *__restrict__ out) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    uint32_t tr = tr0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (tr >= tr1) return;
    uint32_t set = tr >> 3, j = tr & 7;
    uint32_t t = set % T, i = set / T;
    blk_t ctr = make_uint4(8 * i + j, 0, 0, 0);
    out[tr] = aes_encrypt(te, rks + AES_RK_WORDS * t, ctr);
}

// generic-key CTR keystream: out[i] = AES_rk(LE128(first + i))   (util/aes_rng.rs:109-122)
__global__ void k_aes_ctr(const uint32_t *__restrict__ rk, uint64_t first, blk_t *__restrict__ out, uint64_t n) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    __shared__ uint32_t srk[AES_RK_WORDS];
    aes_lds_fill(te);
    if (threadIdx.x < AES_RK_WORDS) srk[threadIdx.x] = rk[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t c = first + i;
        out[i] = aes_encrypt(te, srk, make_uint4((uint32_t)c, (uint32_t)(c >> 32), 0, 0));
    }
}

// mode 0: ECB encrypt, mode 1: x ^ AES_kh(x)
__global__ void k_aes_ecb(const uint32_t *__restrict__ rk, const blk_t *__restrict__ in, blk_t *__restrict__ out,
                          uint64_t n, int mode) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    __shared__ uint32_t srk[AES_RK_WORDS];
    aes_lds_fill(te);
    if (threadIdx.x < AES_RK_WORDS) srk[threadIdx.x] = mode ? c_rk_fixed[AES_KH][threadIdx.x] : rk[threadIdx.x];
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        blk_t x = in[i];
        blk_t e = aes_encrypt(te, srk, x);
        out[i] = mode ? bxor(e, x) : e;
    }
}

// ------------------------------------------------------------------------------------------------
// one GGM level for every tree: child[2j+k] = AES_{k_k}(parent[j]) ^ parent[j], per-level XOR sums
// (pprf.rs:141-194 sender, :388-450 receiver). blockIdx.y = set.
#if GGM_EXP & 2
typedef uint32_t ggm_v4 __attribute__((ext_vector_type(4)));
#define GGM_ST(p, v) __builtin_nontemporal_store((ggm_v4){(v).x, (v).y, (v).z, (v).w}, (ggm_v4 *)(p))
#else
#define GGM_ST(p, v) (*(p) = (v))
#endif
template <bool LAST>
__device__ __forceinline__ void ggm_put(blk_t *__restrict__ out, uint32_t set, uint64_t per_set, uint32_t nsets,
                                        uint64_t cols, uint64_t idx, blk_t c0, blk_t c1) {
    const uint64_t node = idx >> 3;
    const uint32_t t8 = idx & 7;
    if (GGM_EXP & 1) return;
    if (!LAST) {
        blk_t *pout = out + (uint64_t)set * per_set * 2;
        GGM_ST(&pout[(2 * node) * 8 + t8], c0);
        GGM_ST(&pout[(2 * node + 1) * 8 + t8], c1);
    } else {
        // leaves 2*node, 2*node+1 live in the same 16-leaf tile
        const uint64_t leaf = 2 * node;
        const uint64_t tile = set + (uint64_t)nsets * (leaf >> 4);
        if (tile < cols) {
            blk_t *pv = out + tile * 128 + 8 * (leaf & 15) + t8;
            GGM_ST(&pv[0], c0);
            GGM_ST(&pv[8], c1);
        }
    }
}

template <bool LAST>
__global__ void __launch_bounds__(GGM_BLOCK)
    k_ggm_level(const blk_t *__restrict__ in, blk_t *__restrict__ out, uint32_t lvl, uint32_t nsets, uint64_t cols,
                blk_t *__restrict__ sums /* this level: [set][2][8] */, uint32_t set0) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    __shared__ blk_t red[GGM_BLOCK / 64][2][8];
    aes_lds_fill(te);
    const uint32_t set = set0 + blockIdx.y;
    const uint64_t per_set = (uint64_t)8 << lvl;  // parent Blocks of this set
    const blk_t *pin = in + (uint64_t)set * per_set;
    blk_t acc0 = bzero(), acc1 = bzero();
    const uint64_t stride = (uint64_t)gridDim.x * GGM_BLOCK;  // a multiple of 8: a thread keeps its tree%8
    if (per_set % (2 * stride) == 0) {
        // two parents (four independent AES chains) per iteration: the T-table lookups are latency bound otherwise
        for (uint64_t idx = (uint64_t)blockIdx.x * GGM_BLOCK + threadIdx.x; idx < per_set; idx += 2 * stride) {
            const uint64_t idx2 = idx + stride;
            blk_t p = pin[idx], q = pin[idx2];
            blk_t c0 = bxor(aes_encrypt(te, c_rk_fixed[AES_K0], p), p);
            blk_t c1 = bxor(aes_encrypt(te, c_rk_fixed[AES_K1], p), p);
            blk_t e0 = bxor(aes_encrypt(te, c_rk_fixed[AES_K0], q), q);
            blk_t e1 = bxor(aes_encrypt(te, c_rk_fixed[AES_K1], q), q);
            acc0 = bxor(acc0, bxor(c0, e0));
            acc1 = bxor(acc1, bxor(c1, e1));
            ggm_put<LAST>(out, set, per_set, nsets, cols, idx, c0, c1);
            ggm_put<LAST>(out, set, per_set, nsets, cols, idx2, e0, e1);
        }
    } else {
        for (uint64_t idx = (uint64_t)blockIdx.x * GGM_BLOCK + threadIdx.x; idx < per_set; idx += stride) {
            blk_t p = pin[idx];
            blk_t c0 = bxor(aes_encrypt(te, c_rk_fixed[AES_K0], p), p);
            blk_t c1 = bxor(aes_encrypt(te, c_rk_fixed[AES_K1], p), p);
            acc0 = bxor(acc0, c0);
            acc1 = bxor(acc1, c1);
            ggm_put<LAST>(out, set, per_set, nsets, cols, idx, c0, c1);
        }
    }
    // XOR-reduce over lanes with equal tree%8 (lane bits 3..5), then over the block's waves
    for (int off = 8; off < 64; off <<= 1) {
        acc0.x ^= __shfl_xor(acc0.x, off);
        acc0.y ^= __shfl_xor(acc0.y, off);
        acc0.z ^= __shfl_xor(acc0.z, off);
        acc0.w ^= __shfl_xor(acc0.w, off);
        acc1.x ^= __shfl_xor(acc1.x, off);
        acc1.y ^= __shfl_xor(acc1.y, off);
        acc1.z ^= __shfl_xor(acc1.z, off);
        acc1.w ^= __shfl_xor(acc1.w, off);
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 8) {
        red[wave][0][lane] = acc0;
        red[wave][1][lane] = acc1;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        uint32_t keep = threadIdx.x >> 3, t8 = threadIdx.x & 7;
        blk_t s = red[0][keep][t8];
        for (int w = 1; w < GGM_BLOCK / 64; w++) s = bxor(s, red[w][keep][t8]);
        unsigned long long *dst = (unsigned long long *)(sums + ((uint64_t)set * 2 + keep) * 8 + t8);
        atomicXor(dst, ((unsigned long long)s.y << 32) | s.x);
        atomicXor(dst + 1, ((unsigned long long)s.w << 32) | s.z);
    }
}

__device__ __forceinline__ uint64_t active_path(const uint8_t *cb, uint32_t depth);

// ------------------------------------------------------------------------------------------------
// The first levels of every tree, staged in LDS: one workgroup per set (8 trees) expands levels d0 .. d1-1 without
// touching HBM in between -- level l (8 * 2^l Blocks) ping-pongs between two LDS buffers (even l in A, odd l in B),
// only level d1 (<= 9: 4096 Blocks) and the per-level sums are written out. The receiver's repair of the inactive
// child (pprf.rs:456-467, k_recv_fix) happens in LDS between two levels; the repair of level d1 itself is left to
// k_recv_fix. Replaces up to 9 launches of k_ggm_level (+ 8 of k_recv_fix) that were launch-, not AES-bound.
#define GGM_TOP_THREADS 1024
#define GGM_TOP_MAXLVL 9
template <bool RECV>
__global__ void __launch_bounds__(GGM_TOP_THREADS)
    k_ggm_top(const blk_t *__restrict__ in, blk_t *__restrict__ out, uint32_t d0, uint32_t d1, uint32_t nsets,
              blk_t *__restrict__ sums /* [level][set][2][8] */, const blk_t *__restrict__ base,
              const uint8_t *__restrict__ choices, const blk_t *__restrict__ tg_sums, uint32_t depth, uint32_t set0) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    __shared__ blk_t bufA[2048], bufB[1024];
    __shared__ blk_t red[GGM_TOP_THREADS / 64][2][8];
    __shared__ blk_t lsum[2][8];
    aes_lds_fill(te);
    const uint32_t set = set0 + blockIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    {
        const uint32_t n0 = 8u << d0;
        blk_t *b0 = (d0 & 1) ? bufB : bufA;
        for (uint32_t i = threadIdx.x; i < n0; i += GGM_TOP_THREADS) b0[i] = in[(uint64_t)set * n0 + i];
    }
    __syncthreads();
    for (uint32_t d = d0; d < d1; d++) {
        const uint32_t np = 8u << d;  // parent Blocks of this set
        const blk_t *src = (d & 1) ? bufB : bufA;
        blk_t *dst = (d & 1) ? bufA : bufB;
        const bool to_hbm = d + 1 == d1;
        blk_t *gout = out + (uint64_t)set * np * 2;
        blk_t acc0 = bzero(), acc1 = bzero();
        for (uint32_t idx = threadIdx.x; idx < np; idx += GGM_TOP_THREADS) {
            const blk_t p = src[idx];
            const blk_t c0 = bxor(aes_encrypt(te, c_rk_fixed[AES_K0], p), p);
            const blk_t c1 = bxor(aes_encrypt(te, c_rk_fixed[AES_K1], p), p);
            acc0 = bxor(acc0, c0);
            acc1 = bxor(acc1, c1);
            const uint32_t node = idx >> 3, t8 = idx & 7;
            blk_t *o = to_hbm ? gout : dst;
            o[(2 * node) * 8 + t8] = c0;
            o[(2 * node + 1) * 8 + t8] = c1;
        }
        for (int off = 8; off < 64; off <<= 1) {
            acc0.x ^= __shfl_xor(acc0.x, off), acc0.y ^= __shfl_xor(acc0.y, off);
            acc0.z ^= __shfl_xor(acc0.z, off), acc0.w ^= __shfl_xor(acc0.w, off);
            acc1.x ^= __shfl_xor(acc1.x, off), acc1.y ^= __shfl_xor(acc1.y, off);
            acc1.z ^= __shfl_xor(acc1.z, off), acc1.w ^= __shfl_xor(acc1.w, off);
        }
        if (lane < 8) {
            red[wave][0][lane] = acc0;
            red[wave][1][lane] = acc1;
        }
        __syncthreads();
        if (threadIdx.x < 16) {
            const uint32_t keep = threadIdx.x >> 3, t8 = threadIdx.x & 7;
            blk_t sacc = red[0][keep][t8];
            for (int w = 1; w < GGM_TOP_THREADS / 64; w++) sacc = bxor(sacc, red[w][keep][t8]);
            lsum[keep][t8] = sacc;
            sums[(((uint64_t)d * nsets + set) * 2 + keep) * 8 + t8] = sacc;  // the only workgroup of this set: no atomics
        }
        __syncthreads();
        if (RECV && !to_hbm && threadIdx.x < 8) {
            // repair the inactive child of the active parent on level d + 1 (pprf.rs:456-467)
            const uint32_t j = threadIdx.x, tr = set * 8 + j;
            const uint64_t leaf = active_path(choices + (uint64_t)tr * depth, depth);
            const uint64_t inactive = (leaf >> (depth - 1 - d)) ^ 1;
            const uint32_t not_ai = inactive & 1;
            blk_t v = dst[inactive * 8 + j];
            v = bxor(v, tg_sums[(((uint64_t)set * 2 + not_ai) * (depth - 1) + d) * 8 + j]);
            v = bxor(v, lsum[not_ai][j]);
            v = bxor(v, base[(uint64_t)tr * depth + d]);
            dst[inactive * 8 + j] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// sender: mask level sums with the base OTs, build the last-level 4-tuples (pprf.rs:199-254)
__global__ void k_sender_treegrp(const blk_t *__restrict__ sums, const blk_t *__restrict__ base /*[tr*depth+d][2]*/,
                                 blk_t delta, uint32_t parts, uint32_t depth, blk_t *__restrict__ tg_sums,
                                 blk_t *__restrict__ tg_last, uint32_t tr0, uint32_t tr1) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    uint32_t tr = tr0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (tr >= tr1) return;
    const uint32_t nsets = parts >> 3, grp = tr >> 3, j = tr & 7;
    for (uint32_t keep = 0; keep < 2; keep++)
        for (uint32_t d = 0; d + 1 < depth; d++) {
            blk_t s = sums[(((uint64_t)d * nsets + grp) * 2 + keep) * 8 + j];
            tg_sums[(((uint64_t)grp * 2 + keep) * (depth - 1) + d) * 8 + j] =
                bxor(s, base[((uint64_t)tr * depth + d) * 2 + keep]);
        }
    const uint32_t d = depth - 1;
    blk_t s0 = sums[(((uint64_t)d * nsets + grp) * 2 + 0) * 8 + j];
    blk_t s1 = sums[(((uint64_t)d * nsets + grp) * 2 + 1) * 8 + j];
    blk_t o0 = base[((uint64_t)tr * depth + d) * 2 + 0], o1 = base[((uint64_t)tr * depth + d) * 2 + 1];
    blk_t *lo = tg_last + ((uint64_t)grp * 8 + j) * 4;
    lo[0] = bxor(s0, cr_hash(te, o0));
    lo[1] = bxor(bxor(s1, delta), cr_hash(te, bnot(o0)));
    lo[2] = bxor(s1, cr_hash(te, o1));
    lo[3] = bxor(bxor(s0, delta), cr_hash(te, bnot(o1)));
}

// ------------------------------------------------------------------------------------------------
// receiver helpers
__device__ __forceinline__ uint64_t active_path(const uint8_t *cb, uint32_t depth) {  // pprf.rs:701-706
    uint64_t p = 0;
    for (uint32_t i = 0; i < depth; i++) p |= (uint64_t)(1 ^ cb[i]) << (depth - i - 1);
    return p;
}

// level 1 from the first base OT and the first sum (pprf.rs:376-384)
__global__ void k_recv_init(const blk_t *__restrict__ base, const uint8_t *__restrict__ choices,
                            const blk_t *__restrict__ tg_sums, uint32_t parts, uint32_t depth,
                            blk_t *__restrict__ lvl1, uint32_t tr0, uint32_t tr1) {
    uint32_t tr = tr0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (tr >= tr1) return;
    uint32_t grp = tr >> 3, j = tr & 7;
    uint32_t not_ai = choices[(uint64_t)tr * depth];
    blk_t v = bxor(base[(uint64_t)tr * depth], tg_sums[(((uint64_t)grp * 2 + not_ai) * (depth - 1)) * 8 + j]);
    lvl1[((uint64_t)grp * 2 + not_ai) * 8 + j] = v;
    lvl1[((uint64_t)grp * 2 + (not_ai ^ 1)) * 8 + j] = bzero();
}

// repair the inactive child of the active parent on level d+1 (pprf.rs:456-467)
__global__ void k_recv_fix(const blk_t *__restrict__ base, const uint8_t *__restrict__ choices,
                           const blk_t *__restrict__ tg_sums, const blk_t *__restrict__ my_sums /*level d*/,
                           uint32_t parts, uint32_t depth, uint32_t d, blk_t *__restrict__ lvl /*level d+1*/, uint32_t tr0,
                           uint32_t tr1) {
    uint32_t tr = tr0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (tr >= tr1) return;
    uint32_t grp = tr >> 3, j = tr & 7;
    uint64_t leaf = active_path(choices + (uint64_t)tr * depth, depth);
    uint64_t active = leaf >> (depth - 1 - d);
    uint64_t inactive = active ^ 1;
    uint32_t not_ai = inactive & 1;
    blk_t *p = lvl + (((uint64_t)grp << (d + 1)) + inactive) * 8 + j;
    blk_t v = *p;
    v = bxor(v, tg_sums[(((uint64_t)grp * 2 + not_ai) * (depth - 1) + d) * 8 + j]);
    v = bxor(v, my_sums[((uint64_t)grp * 2 + not_ai) * 8 + j]);
    v = bxor(v, base[(uint64_t)tr * depth + d]);
    *p = v;
}

// last level: fix both children of the active parent inside V (pprf.rs:474-535)
__global__ void k_recv_last(const blk_t *__restrict__ base, const uint8_t *__restrict__ choices,
                            const blk_t *__restrict__ tg_last, const blk_t *__restrict__ my_sums /*level depth-1*/,
                            uint32_t parts, uint32_t depth, uint64_t cols, blk_t *__restrict__ V, uint32_t tr0, uint32_t tr1) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    uint32_t tr = tr0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (tr >= tr1) return;
    const uint32_t nsets = parts >> 3, grp = tr >> 3, j = tr & 7;
    uint64_t active = active_path(choices + (uint64_t)tr * depth, depth);
    uint64_t inactive = active ^ 1;
    uint32_t not_ai = inactive & 1;
    blk_t o = base[(uint64_t)tr * depth + depth - 1];
    blk_t ot0 = bxor(tg_last[((uint64_t)grp * 8 + j) * 4 + 2 * not_ai + 0], cr_hash(te, o));
    blk_t ot1 = bxor(tg_last[((uint64_t)grp * 8 + j) * 4 + 2 * not_ai + 1], cr_hash(te, bnot(o)));
    uint64_t tile = grp + (uint64_t)nsets * (active >> 4);
    if (tile >= cols) return;  // pprf.rs:509-519 guard; unreachable for sampled choices
    blk_t *pi = V + tile * 128 + 8 * (inactive & 15) + j;
    blk_t *pa = V + tile * 128 + 8 * (active & 15) + j;
    blk_t inactive_sum = bxor(my_sums[((uint64_t)grp * 2 + not_ai) * 8 + j], *pi);
    blk_t active_sum = bxor(my_sums[((uint64_t)grp * 2 + (not_ai ^ 1)) * 8 + j], *pa);
    *pi = bxor(ot0, inactive_sum);
    *pa = bxor(ot1, active_sum);
}

// ------------------------------------------------------------------------------------------------
// 128x128 bit-tile transposes (util/transpose.rs:53-109 semantics, LSB first):
// out_tile[r] bit j = in_tile[j] bit r. One 128-thread block per tile.
__global__ void __launch_bounds__(128)
    k_transpose_tiles(const blk_t *__restrict__ in, uint64_t in_estride, uint64_t in_tstride, blk_t *__restrict__ out,
                      uint64_t out_estride, uint64_t out_tstride, uint64_t ntiles) {
    __shared__ uint32_t xch[512];
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const uint32_t r = threadIdx.x;
        const blk_t v = in[t * in_tstride + (uint64_t)r * in_estride];
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
        bm_transpose128(w, xch, r);  // registers + a 4 x 4 word exchange through LDS (bitmat_dev.h)
        out[t * out_tstride + (uint64_t)r * out_estride] = make_uint4(w[0], w[1], w[2], w[3]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// hashes (silent_ot/mod.rs:275-287, 470-482) and MT bits (gmw/src/mul_triple/silent_ot.rs:93-111)
__global__ void k_sender_hash(const blk_t *__restrict__ B, blk_t delta, uint64_t n, blk_t *__restrict__ out) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    blk_t d = delta;
    d.x &= ~1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        blk_t m = B[i];
        m.x &= ~1u;
        out[2 * i] = cr_hash(te, m);
        out[2 * i + 1] = cr_hash(te, bxor(m, d));
    }
}

__global__ void k_receiver_hash(const blk_t *__restrict__ A, uint64_t n, blk_t *__restrict__ out,
                                uint64_t *__restrict__ choice_words) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad;
         i += (uint64_t)gridDim.x * blockDim.x) {
        blk_t m = i < n ? A[i] : bzero();
        unsigned long long bal = __ballot(m.x & 1u);
        m.x &= ~1u;
        if (i < n) out[i] = cr_hash(te, m);
        if ((threadIdx.x & 63) == 0) choice_words[i >> 6] = bal;
    }
}

// a = lsb(A); b = lsb(m0)^lsb(m1); v = lsb(m0); u = lsb(H(A&~1)); c = a&b ^ u ^ v
__global__ void k_mt_bits(const blk_t *__restrict__ B, const blk_t *__restrict__ A, blk_t delta, uint64_t n,
                          uint64_t *__restrict__ wa, uint64_t *__restrict__ wb, uint64_t *__restrict__ wc) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    blk_t d = delta;
    d.x &= ~1u;
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t a = 0, b = 0, c = 0;
        if (i < n) {
            blk_t mb = B[i], ma = A[i];
            a = ma.x & 1u;
            mb.x &= ~1u;
            ma.x &= ~1u;
            uint32_t m0 = cr_hash(te, mb).x & 1u;
            uint32_t m1 = cr_hash(te, bxor(mb, d)).x & 1u;
            uint32_t u = cr_hash(te, ma).x & 1u;
            b = m0 ^ m1;
            c = (a & b) ^ u ^ m0;
        }
        unsigned long long ba = __ballot(a), bb = __ballot(b), bc = __ballot(c);
        if ((threadIdx.x & 63) == 0) {
            wa[i >> 6] = ba;
            wb[i >> 6] = bb;
            wc[i >> 6] = bc;
        }
    }
}

// The same bits by direction, for a party that runs its two directions on two GPUs (role split, shard.hip / dist.py): the
// reference computes them on two thread pools and joins (gmw/src/mul_triple/silent_ot.rs:87-111). Sender direction:
// b = lsb(m0) ^ lsb(m1), v = lsb(m0) of its own messages; receiver direction: a = lsb(A), u = lsb(H(A & ~1)); the join is
// c = a & b ^ u ^ v, word-wise.
__global__ void k_mt_role_bits(const blk_t *__restrict__ E, int sender, blk_t delta, uint64_t n, uint64_t *__restrict__ w0,
                               uint64_t *__restrict__ w1) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    blk_t d = delta;
    d.x &= ~1u;
    const uint64_t n_pad = (n + 63) & ~63ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t x0 = 0, x1 = 0;
        if (i < n) {
            blk_t m = E[i];
            const uint32_t lsb = m.x & 1u;
            m.x &= ~1u;
            const uint32_t h0 = cr_hash(te, m).x & 1u;
            if (sender) {
                x0 = h0 ^ (cr_hash(te, bxor(m, d)).x & 1u);  // b
                x1 = h0;                                      // v
            } else {
                x0 = lsb;  // a
                x1 = h0;   // u
            }
        }
        const unsigned long long b0 = __ballot(x0), b1 = __ballot(x1);
        if ((threadIdx.x & 63) == 0) {
            w0[i >> 6] = b0;
            w1[i >> 6] = b1;
        }
    }
}

// ================================================================================================
// host side
// ================================================================================================
static uint8_t sbox_host(uint8_t x) {
    // derive S-box from TE0 byte 1 (TE0 = (2S, S, S, 3S))
    static const uint32_t te0[256] = {
#include "aes_te0.inc"
    };
    return (uint8_t)(te0[x] >> 8);
}

void aes_expand_key_host(const uint8_t key[16], aes_rk *out) {
    static const uint8_t rcon[10] = {0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40, 0x80, 0x1b, 0x36};
    uint8_t rk[176];
    memcpy(rk, key, 16);
    for (int i = 4; i < 44; i++) {
        uint8_t t[4];
        memcpy(t, rk + 4 * (i - 1), 4);
        if (i % 4 == 0) {
            uint8_t u0 = t[0];
            t[0] = sbox_host(t[1]) ^ rcon[i / 4 - 1];
            t[1] = sbox_host(t[2]);
            t[2] = sbox_host(t[3]);
            t[3] = sbox_host(u0);
        }
        for (int k = 0; k < 4; k++) rk[4 * i + k] = rk[4 * (i - 4) + k] ^ t[k];
    }
    memcpy(out->w, rk, 176);
    for (int i = 0; i < 44; i++) out->w[44 + i] = (out->w[i] >> 8) | (out->w[i] << 24);
}

static inline unsigned grid_for(uint64_t n, unsigned block, unsigned cap = 4096) {
    uint64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

static int launch_levels(sot_ctx *c, bool receiver, blk_t *V) {
    const uint32_t depth = (uint32_t)c->conf.depth, nsets = (uint32_t)c->nsets;
    const uint32_t set0 = c->ggm_set0, nl = c->ggm_nsets;  // this context's tree sets (all of them on one GPU)
    const uint32_t tr0 = 8 * set0, tr1 = 8 * (set0 + nl);
    const dim3 gt((8 * nl + 127) / 128), bt(128);
    // sender: levels 0..depth-1 from the roots in lvl_a; receiver: levels 1..depth-1 from level 1 in lvl_b
    blk_t *cur = receiver ? c->lvl_b : c->lvl_a;
    blk_t *nxt = receiver ? c->lvl_a : c->lvl_b;
    uint32_t dstart = receiver ? 1 : 0;
    // levels dstart .. d1-1 in LDS (k_ggm_top); level d1 lands in nxt. The leaf level always goes through k_ggm_level<true>.
    const uint32_t d1 = depth - 1 < GGM_TOP_MAXLVL ? depth - 1 : GGM_TOP_MAXLVL;
    if (d1 > dstart && !(c->opts & SOT_OPT_NO_GGM_TOP)) {
        if (receiver) {
            SOT_LAUNCH(c, KID_GGM_LEVEL, k_ggm_top<true>, dim3(nl), dim3(GGM_TOP_THREADS), cur, nxt, dstart, d1, nsets,
                       c->sums, c->d_base, c->d_choices, c->d_tg_sums, depth, set0);
            SOT_LAUNCH(c, KID_RECV_FIX, k_recv_fix, gt, bt, c->d_base, c->d_choices, c->d_tg_sums,
                       c->sums + (uint64_t)(d1 - 1) * nsets * 16, (uint32_t)c->conf.parts, depth, d1 - 1, nxt, tr0, tr1);
        } else {
            SOT_LAUNCH(c, KID_GGM_LEVEL, k_ggm_top<false>, dim3(nl), dim3(GGM_TOP_THREADS), cur, nxt, dstart, d1, nsets,
                       c->sums, (const blk_t *)nullptr, (const uint8_t *)nullptr, (const blk_t *)nullptr, depth, set0);
        }
        blk_t *t = cur;
        cur = nxt;
        nxt = t;
        dstart = d1;
    }
    for (uint32_t d = dstart; d < depth; d++) {
        uint64_t per_set = (uint64_t)8 << d;
        // interior levels: few, longer-lived workgroups (each fills the 64 KiB AES table once; 15 sets x 64 ~ 4 per CU);
        // the leaf level has enough work per workgroup and balances better with many. A shard with fewer sets takes
        // proportionally more workgroups per set.
        unsigned cap = d + 1 < depth ? 64 : 256;
        if (nl < nsets) {
            // ... rounded up to a power of two: the grid stride must divide the level for the kernel's four-chain path
            // (per_set % (2 * stride) == 0); 64 * 15 / 2 = 480 workgroups sent every level of an 8-rank shard down the
            // two-chain path
            unsigned want = cap * (unsigned)nsets / nl, p2 = cap;
            while (p2 < want) p2 <<= 1;
            cap = p2;
        }
        unsigned gx = grid_for(per_set, GGM_BLOCK, cap);
        // a shard's level holds nl / nsets of the work: with the workgroup count of a whole level each workgroup would fill
        // its 64 KiB AES table for one or two parents per thread. Halve the grid while more than two workgroups per CU
        // would each see fewer than 16 parents per thread.
        if (nl < nsets)
            while (gx > 1 && (uint64_t)gx * nl > 2 * c->num_cus && per_set / ((uint64_t)gx * GGM_BLOCK) < 16) gx >>= 1;
        dim3 grid(gx, nl);
        blk_t *lsum = c->sums + (uint64_t)d * nsets * 16;
        if (d + 1 < depth) {
            SOT_LAUNCH(c, KID_GGM_LEVEL, k_ggm_level<false>, grid, dim3(GGM_BLOCK), cur, nxt, d, nsets,
                               c->conf.cols, lsum, set0);
            if (receiver)
                SOT_LAUNCH(c, KID_RECV_FIX, k_recv_fix, gt, bt, c->d_base, c->d_choices, c->d_tg_sums, lsum,
                           (uint32_t)c->conf.parts, depth, d, nxt, tr0, tr1);
            blk_t *t = cur;
            cur = nxt;
            nxt = t;
        } else {
            SOT_LAUNCH(c, KID_GGM_LEAF, k_ggm_level<true>, grid, dim3(GGM_BLOCK), cur, V, d, nsets, c->conf.cols,
                               lsum, set0);
        }
    }
    SOT_HIP(hipGetLastError());
    return SOT_OK;
}

int ggm_sender_expand(sot_ctx *c, const void *base_ots, const void *delta, const void *seed, uint64_t T,
                      void *tg_sums_out, void *tg_last_out) {
    const sot_conf &cf = c->conf;
    if (T == 0 || T > 4096) {
        sot_set_error("num_threads must be in 1..4096");
        return SOT_EINVAL;
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint32_t tr0 = 8 * c->ggm_set0, tr1 = 8 * (c->ggm_set0 + c->ggm_nsets);
    const dim3 gt((tr1 - tr0 + 127) / 128), bt(128);
    // root PRG keys: seed ^ t (pprf.rs:115, Block::from(usize) is the low 64 bits)
    uint64_t Teff = T < c->nsets ? T : c->nsets;
    std::vector<aes_rk> rks(Teff);
    for (uint64_t t = 0; t < Teff; t++) {
        uint8_t key[16];
        memcpy(key, seed, 16);
        uint64_t lo;
        memcpy(&lo, key, 8);
        lo ^= t;
        memcpy(key, &lo, 8);
        aes_expand_key_host(key, &rks[t]);
    }
    if (Teff > c->rks_cap) {  // cannot happen: sot_ctx_create sized d_rks for one worker per tree group
        sot_set_error("internal: root key buffer too small");
        return SOT_EINVAL;
    }
    SOT_HIP(hipMemcpyAsync(c->d_rks, rks.data(), Teff * sizeof(aes_rk), hipMemcpyHostToDevice, c->stream));
    SOT_HIP(hipMemcpyAsync(c->d_base, base_ots, cf.base_ot_count * 32, hipMemcpyHostToDevice, c->stream));
    SOT_HIP(hipMemsetAsync(c->sums, 0, cf.depth * c->nsets * 16 * sizeof(blk_t), c->stream));
    blk_t *V = c->role[SOT_ROLE_SENDER].V;
    SOT_LAUNCH(c, KID_ROOTS, k_roots, gt, bt, c->d_rks, (uint32_t)T, tr0, tr1, c->lvl_a);
    SOT_TRY(launch_levels(c, false, V));
    blk_t d;
    memcpy(&d, delta, 16);
    SOT_LAUNCH(c, KID_SENDER_TG, k_sender_treegrp, gt, bt, c->sums, c->d_base, d, (uint32_t)cf.parts, (uint32_t)cf.depth,
               c->d_tg_sums, c->d_tg_last, tr0, tr1);
    SOT_HIP(hipGetLastError());
    // the TreeGrps of this context's tree sets (a contiguous range of groups in both flat arrays); a sharded context
    // leaves the other groups' entries of the caller's arrays untouched
    const uint64_t sb1 = 2 * (cf.depth - 1) * 8 * 16, lb1 = 8 * 4 * 16;  // bytes per group
    if (tg_sums_out)
        SOT_HIP(hipMemcpyAsync((char *)tg_sums_out + c->ggm_set0 * sb1, (const char *)c->d_tg_sums + c->ggm_set0 * sb1,
                               c->ggm_nsets * sb1, hipMemcpyDeviceToHost, c->stream));
    if (tg_last_out)
        SOT_HIP(hipMemcpyAsync((char *)tg_last_out + c->ggm_set0 * lb1, (const char *)c->d_tg_last + c->ggm_set0 * lb1,
                               c->ggm_nsets * lb1, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    c->role[SOT_ROLE_SENDER].expanded = true;
    c->role[SOT_ROLE_SENDER].encoded = false;
    c->sh.slots_ready = c->sh.enc_slots_ready = false;
    c->mt_valid = false;
    return SOT_OK;
}

int ggm_receiver_expand(sot_ctx *c, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                        const void *tg_last) {
    const sot_conf &cf = c->conf;
    SOT_HIP(hipSetDevice(c->device));
    for (uint64_t i = 0; i < cf.base_ot_count; i++)
        if (choices[i] > 1) {
            sot_set_error("choices must be 0/1 bytes");
            return SOT_EINVAL;
        }
    const uint32_t tr0 = 8 * c->ggm_set0, tr1 = 8 * (c->ggm_set0 + c->ggm_nsets);
    const dim3 gt((tr1 - tr0 + 127) / 128), bt(128);
    SOT_TRY(stage_h2d(c, c->d_base, base_ots, cf.base_ot_count * 16));
    SOT_TRY(stage_h2d(c, c->d_choices, choices, cf.base_ot_count));
    SOT_TRY(stage_h2d(c, c->d_tg_sums, tg_sums, sot_treegrp_sums_blocks(&cf) * 16));
    SOT_TRY(stage_h2d(c, c->d_tg_last, tg_last, sot_treegrp_last_blocks(&cf) * 16));
    SOT_HIP(hipMemsetAsync(c->sums, 0, cf.depth * c->nsets * 16 * sizeof(blk_t), c->stream));
    blk_t *V = c->role[SOT_ROLE_RECEIVER].V;
    SOT_LAUNCH(c, KID_RECV_INIT, k_recv_init, gt, bt, c->d_base, c->d_choices, c->d_tg_sums, (uint32_t)cf.parts,
                       (uint32_t)cf.depth, c->lvl_b, tr0, tr1);
    SOT_TRY(launch_levels(c, true, V));
    SOT_LAUNCH(c, KID_RECV_LAST, k_recv_last, gt, bt, c->d_base, c->d_choices, c->d_tg_last,
                       c->sums + (cf.depth - 1) * c->nsets * 16, (uint32_t)cf.parts, (uint32_t)cf.depth, cf.cols, V, tr0, tr1);
    SOT_HIP(hipGetLastError());
    SOT_SYNC(c);
    c->role[SOT_ROLE_RECEIVER].expanded = true;
    c->role[SOT_ROLE_RECEIVER].encoded = false;
    c->sh.slots_ready = c->sh.enc_slots_ready = false;
    c->mt_valid = false;
    return SOT_OK;
}

// the staging buffer is allocated once, in sot_ctx_create, for the largest user: no hipMalloc on any stage call
static int ensure_tmp(sot_ctx *c, uint64_t bytes) {
    if (c->tmp_bytes >= bytes) return SOT_OK;
    sot_set_error("internal: staging buffer of %llu bytes, need %llu", (unsigned long long)c->tmp_bytes,
                  (unsigned long long)bytes);
    return SOT_EINVAL;
}

int ggm_download_rT(sot_ctx *c, int role, void *out) {
    const sot_conf &cf = c->conf;
    if (!c->role[role].expanded) {
        sot_set_error("download_rT before expand");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    uint64_t bytes = 128 * cf.cols * 16;
    SOT_TRY(ensure_tmp(c, bytes));
    SOT_LAUNCH(c, KID_TRANSPOSE, k_transpose_tiles, dim3(grid_for(cf.cols, 1, 8192)), dim3(128), c->role[role].V,
                       (uint64_t)1, (uint64_t)128, (blk_t *)c->d_tmp, cf.cols, (uint64_t)1, cf.cols);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipMemcpyAsync(out, c->d_tmp, bytes, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    return SOT_OK;
}

int ggm_upload_rT(sot_ctx *c, int role, const void *rT) {
    const sot_conf &cf = c->conf;
    SOT_HIP(hipSetDevice(c->device));
    uint64_t bytes = 128 * cf.cols * 16;
    SOT_TRY(ensure_tmp(c, bytes));
    SOT_HIP(hipMemcpyAsync(c->d_tmp, rT, bytes, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_transpose_tiles, dim3(grid_for(cf.cols, 1, 8192)), dim3(128), 0, c->stream,
                       (const blk_t *)c->d_tmp, cf.cols, (uint64_t)1, c->role[role].V, (uint64_t)1, (uint64_t)128,
                       cf.cols);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));
    c->role[role].expanded = true;
    c->role[role].encoded = false;
    c->mt_valid = false;  // the MT store no longer belongs to this context's leaves
    return SOT_OK;
}

int hash_sender(sot_ctx *c, const void *delta, void *msgs_out, blk_t *dst_dev) {
    const sot_conf &cf = c->conf;
    if (!c->role[SOT_ROLE_SENDER].encoded) {
        sot_set_error("sender_hash before qc_encode");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    if (!dst_dev) {
        SOT_TRY(ensure_tmp(c, c->n_out * 32));
        dst_dev = (blk_t *)c->d_tmp;
    }
    blk_t d;
    memcpy(&d, delta, 16);
    SOT_LAUNCH(c, KID_SENDER_HASH, k_sender_hash, dim3(grid_for(c->n_out, HASH_BLOCK, 2048)), dim3(HASH_BLOCK),
                       c->role[SOT_ROLE_SENDER].enc, d, c->n_out, dst_dev);
    SOT_HIP(hipGetLastError());
    if (msgs_out) SOT_HIP(hipMemcpyAsync(msgs_out, dst_dev, c->n_out * 32, hipMemcpyDeviceToHost, c->stream));
    if (msgs_out) SOT_HIP(hipStreamSynchronize(c->stream));
    else SOT_SYNC(c);
    return SOT_OK;
}

int hash_receiver(sot_ctx *c, void *msgs_out, uint64_t *choices_out, blk_t *dst_dev) {
    const sot_conf &cf = c->conf;
    if (!c->role[SOT_ROLE_RECEIVER].encoded) {
        sot_set_error("receiver_hash before qc_encode");
        return SOT_ESTATE;
    }
    if (c->recv_unpacked) {  // Receiver::hash reads the choice from bit 0 of A (silent_ot/mod.rs:470-477)
        sot_set_error("receiver_hash needs ChoiceBitPacking::True; the last encode was unpacked");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    if (!dst_dev) {
        SOT_TRY(ensure_tmp(c, c->n_out * 32));
        dst_dev = (blk_t *)c->d_tmp;
    }
    uint64_t nw = (c->n_out + 63) / 64;
    SOT_LAUNCH(c, KID_RECEIVER_HASH, k_receiver_hash, dim3(grid_for(c->n_out, HASH_BLOCK, 2048)), dim3(HASH_BLOCK),
                       c->role[SOT_ROLE_RECEIVER].enc, c->n_out, dst_dev, c->d_chw);
    SOT_HIP(hipGetLastError());
    if (msgs_out) SOT_HIP(hipMemcpyAsync(msgs_out, dst_dev, c->n_out * 16, hipMemcpyDeviceToHost, c->stream));
    if (choices_out) SOT_HIP(hipMemcpyAsync(choices_out, c->d_chw, nw * 8, hipMemcpyDeviceToHost, c->stream));
    if (msgs_out || choices_out) SOT_HIP(hipStreamSynchronize(c->stream));
    else SOT_SYNC(c);
    return SOT_OK;
}

// dst_device: 2 * ceil(n_out / 64) words: b | v (sender) or a | u (receiver)
int hash_mt_role_bits(sot_ctx *c, int role, const void *delta, uint64_t *dst_device) {
    if (!c->role[role].encoded) {
        sot_set_error("mt_role_bits: this direction is not encoded yet");
        return SOT_ESTATE;
    }
    if (role == SOT_ROLE_RECEIVER && c->recv_unpacked) {
        sot_set_error("mt_role_bits needs ChoiceBitPacking::True; the receiver's last encode was unpacked");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nw = (c->n_out + 63) / 64;
    blk_t d = make_uint4(0, 0, 0, 0);
    if (role == SOT_ROLE_SENDER) memcpy(&d, delta, 16);
    SOT_LAUNCH(c, KID_MT_BITS, k_mt_role_bits, dim3(grid_for(c->n_out, HASH_BLOCK, 2048)), dim3(HASH_BLOCK), c->role[role].enc,
               role == SOT_ROLE_SENDER ? 1 : 0, d, c->n_out, dst_device, dst_device + nw);
    SOT_HIP(hipGetLastError());
    SOT_SYNC(c);
    return SOT_OK;
}

int hash_mt_bits(sot_ctx *c, const void *delta, uint64_t *a, uint64_t *b, uint64_t *cc) {
    const sot_conf &cf = c->conf;
    if (!c->role[0].encoded || !c->role[1].encoded) {
        sot_set_error("mt_bits needs both the sender and the receiver instance encoded");
        return SOT_ESTATE;
    }
    if (c->recv_unpacked) {
        sot_set_error("mt_bits needs ChoiceBitPacking::True; the receiver's last encode was unpacked");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    uint64_t nw = (c->n_out + 63) / 64;
    blk_t d;
    memcpy(&d, delta, 16);
    SOT_LAUNCH(c, KID_MT_BITS, k_mt_bits, dim3(grid_for(c->n_out, HASH_BLOCK, 2048)), dim3(HASH_BLOCK), c->role[0].enc,
                       c->role[1].enc, d, c->n_out, c->d_mt, c->d_mt + nw, c->d_mt + 2 * nw);
    SOT_HIP(hipGetLastError());
    c->mt_valid = true;
    c->mt_rem = c->n_out;
    if (a) SOT_HIP(hipMemcpyAsync(a, c->d_mt, nw * 8, hipMemcpyDeviceToHost, c->stream));
    if (b) SOT_HIP(hipMemcpyAsync(b, c->d_mt + nw, nw * 8, hipMemcpyDeviceToHost, c->stream));
    if (cc) SOT_HIP(hipMemcpyAsync(cc, c->d_mt + 2 * nw, nw * 8, hipMemcpyDeviceToHost, c->stream));
    if (a || b || cc) SOT_HIP(hipStreamSynchronize(c->stream));
    else SOT_SYNC(c);
    return SOT_OK;
}

// one cleanup path for the stateless helpers: every allocation is freed and every HIP call checked
struct dev_bufs {
    void *p[4] = {nullptr, nullptr, nullptr, nullptr};
    int n = 0;
    int alloc(void **out, uint64_t bytes) {
        *out = nullptr;
        if (hipMalloc(out, bytes ? bytes : 16) != hipSuccess) {
            sot_set_error("hipMalloc(%llu bytes) failed", (unsigned long long)bytes);
            return SOT_ENOMEM;
        }
        p[n++] = *out;
        return SOT_OK;
    }
    ~dev_bufs() {
        for (int i = 0; i < n; i++) (void)hipFree(p[i]);
    }
};

int util_transpose128(int device, const void *in, void *out, uint64_t ntiles) {
    SOT_HIP(hipSetDevice(device));
    if (ntiles == 0) return SOT_OK;
    dev_bufs bufs;
    blk_t *din = nullptr, *dout = nullptr;
    const uint64_t bytes = ntiles * 2048;
    SOT_TRY(bufs.alloc((void **)&din, bytes));
    SOT_TRY(bufs.alloc((void **)&dout, bytes));
    SOT_HIP(hipMemcpy(din, in, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_transpose_tiles, dim3(grid_for(ntiles, 1, 8192)), dim3(128), 0, 0, din, (uint64_t)1,
                       (uint64_t)128, dout, (uint64_t)1, (uint64_t)128, ntiles);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipMemcpy(out, dout, bytes, hipMemcpyDeviceToHost));
    return SOT_OK;
}

int util_aes(int device, const void *key16, const void *in, void *out, uint64_t n, int cr) {
    SOT_HIP(hipSetDevice(device));
    if (n == 0) return SOT_OK;
    aes_rk rk;
    memset(&rk, 0, sizeof(rk));
    if (!cr) aes_expand_key_host((const uint8_t *)key16, &rk);
    dev_bufs bufs;
    blk_t *din = nullptr, *dout = nullptr;
    uint32_t *drk = nullptr;
    SOT_TRY(bufs.alloc((void **)&din, n * 16));
    SOT_TRY(bufs.alloc((void **)&dout, n * 16));
    SOT_TRY(bufs.alloc((void **)&drk, sizeof(rk)));
    SOT_HIP(hipMemcpy(din, in, n * 16, hipMemcpyHostToDevice));
    SOT_HIP(hipMemcpy(drk, &rk, sizeof(rk), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_aes_ecb, dim3(grid_for(n, 256, 2048)), dim3(256), 0, 0, drk, din, dout, n, cr);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipMemcpy(out, dout, n * 16, hipMemcpyDeviceToHost));
    return SOT_OK;
}

int util_aes_ctr_device(sot_ctx *c, const uint8_t key16[16], uint64_t first, blk_t *d_out, uint64_t nblocks) {
    aes_rk rk;
    aes_expand_key_host(key16, &rk);
    dev_bufs bufs;
    uint32_t *drk = nullptr;
    SOT_TRY(bufs.alloc((void **)&drk, sizeof(rk)));
    SOT_HIP(hipMemcpyAsync(drk, &rk, sizeof(rk), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_aes_ctr, dim3(grid_for(nblocks, 256, 2048)), dim3(256), 0, c->stream, drk, first, d_out,
                       nblocks);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));  // before bufs frees drk
    return SOT_OK;
}
