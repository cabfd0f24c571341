// ALSZ13 OT extension on gfx950: the local work of zappot::ot_ext::{Sender, Receiver}
// (crates/zappot/src/ot_ext/mod.rs:48-142, :156-237) after the 128 base OTs exist.
//
//   receiver (:193-235): t_row[r] = PRG(s0_r), u_row[r] = PRG(s1_r) ^ t_row[r] ^ choices  (r < 128, count/8 bytes each),
//                        T = transpose(t_mat, 128, count), out[i] = H(T_i)
//   sender   (:91-140) : v_row[r] = PRG(seed_r) ^ (base_choice_r ? u_row[r] : 0), Q = transpose(v_mat, 128, count),
//                        out[i] = [H(Q_i), H(Q_i ^ delta)], delta = the 128 base choices as a Block
//   PRG = AesRng (util/aes_rng.rs:101-136): block j of a row = AES_seed(LE128(j)); H = cr_hash_block (aes_hash.rs:26-30);
//   transpose = util/transpose.rs:5-14 (bit r of output Block i = bit i of row r, Lsb0 bytes).
//
// One tile = keystream block j of all 128 rows = OTs 128 j .. 128 j + 127: thread r encrypts counter j under row r's
// key (the 128 or 256 expanded keys sit in LDS, 44 words each), the 128 x 128 bit tile is transposed through LDS and
// every thread hashes one OT. Workgroups of 1024 threads take 8 tiles at a time.
#include <string.h>

#include <mutex>
#include <vector>

#include "aes_dev.h"
#include "bitmat_dev.h"
#include "sot_internal.h"

#define AZ_THREADS 1024
#define AZ_TILES (AZ_THREADS / 64)  // one wavefront per 128-OT tile
#ifndef AZ_GRID
#define AZ_GRID 256  // workgroups: one per CU (112 KiB of LDS each), every one fills its tables once
#endif

// words per row of the per-lane round keys in LDS: 44 (sender: one key per base OT) or 2 x 44 + 4 (receiver: the pad makes
// the row stride an odd number of 16-byte pieces, so the ds_read_b128 of 8 consecutive lanes fall into different banks)
#define AZ_KSTRIDE(SENDER) ((SENDER) ? 44 : 92)

// A wavefront owns a whole tile (column j of the 128 x nj Block matrix): lane l computes rows l and l + 64 -- two
// (sender) or four (receiver) independent AES chains per lane --, the 128 x 128 bit transpose stays in the wave's
// registers (bm_transpose128_wave) and the lane hashes OTs 128 j + l and 128 j + l + 64. No barrier inside the loop: the
// sixteen waves of the workgroup drift apart and overlap each other's lookup, transpose and store phases.
template <bool SENDER>
__global__ void __launch_bounds__(AZ_THREADS)
    k_alsz(const uint32_t *__restrict__ seeds /* the base-OT Blocks = AES keys; SENDER: [128][4] words, else [128][2][4] */,
           const blk_t *__restrict__ choices,
           blk_t delta, const uint8_t *__restrict__ base_choice /* SENDER: [128] 0/1 */, blk_t *__restrict__ U /* [128][nj] */,
           uint64_t nj, uint64_t count, blk_t *__restrict__ out) {
    constexpr uint32_t KS = AZ_KSTRIDE(SENDER);
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    __shared__ __align__(16) uint32_t keys[128 * KS];
    __shared__ uint32_t s_ctr;
    if (threadIdx.x == 0) s_ctr = 0;
    aes_lds_fill(te);
    __syncthreads();
    // FIPS-197 key expansion of the 128 / 256 base-OT keys, one key per thread, S-box = byte 1 of the table's TE0 replicas
    // (TE0 = (2S, S, S, 3S)); on the host it was 40 us of every call
    if (threadIdx.x < 128 * (SENDER ? 1 : 2)) {
        uint32_t *kk = keys + (SENDER ? threadIdx.x * KS : (threadIdx.x >> 1) * KS + (threadIdx.x & 1u) * 44);
        const uint4 k4 = *(const uint4 *)(seeds + 4 * threadIdx.x);
        uint32_t w0 = k4.x, w1 = k4.y, w2 = k4.z, w3 = k4.w, rcon = 1;
        kk[0] = w0, kk[1] = w1, kk[2] = w2, kk[3] = w3;
#pragma unroll 1
        for (int r = 1; r <= 10; r++) {
            auto S = [&](uint32_t x) { return (te[0][(x & 255u) * 64] >> 8) & 255u; };
            // SubWord(RotWord(w3)) ^ rcon, bytes little-endian in the word
            w0 ^= (S(w3 >> 8) | (S(w3 >> 16) << 8) | (S(w3 >> 24) << 16) | (S(w3) << 24)) ^ rcon;
            w1 ^= w0, w2 ^= w1, w3 ^= w2;
            kk[4 * r] = w0, kk[4 * r + 1] = w1, kk[4 * r + 2] = w2, kk[4 * r + 3] = w3;
            rcon = (rcon << 1) ^ ((rcon & 0x80u) ? 0x11bu : 0u);
        }
    }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *ka = keys + lane * KS, *kb = keys + (lane + 64) * KS;
    const uint32_t rbit_a = SENDER ? base_choice[lane] : 0, rbit_b = SENDER ? base_choice[lane + 64] : 0;
    blk_t *Ua = U + (uint64_t)lane * nj, *Ub = U + (uint64_t)(lane + 64) * nj;
    // A wave DRAWS its tiles from the workgroup's share (tile k of workgroup b is column (k / 16) * 16 gridDim + 16 b + k % 16;
    // k comes from a counter in LDS, the first one is the wave's own index): the CU serves its oldest waves first, so with
    // equal fixed shares they finish early and the youngest run the tail of the kernel with too few waves beside them to
    // keep the LDS pipe busy (tools/micro/cu_spread.hip). One counter for the whole grid would serialise 2^17 atomics on
    // one address (1.5 ms instead of 0.7); between CUs the spread is 1 %.
    const uint64_t nfix = (uint64_t)gridDim.x * AZ_TILES;
    auto column = [&](uint32_t k) { return (uint64_t)(k >> 4) * nfix + (uint64_t)blockIdx.x * AZ_TILES + (k & 15u); };
    uint64_t j = column(wave);
    for (; j < nj; ) {
        uint32_t drawn = 0;
        if (lane == 0) drawn = atomicAdd(&s_ctr, 1u);
        const blk_t ctr = make_uint4((uint32_t)j, (uint32_t)(j >> 32), 0, 0);
        blk_t ta, tb;
        // receiver: rows l and l + 64 one after the other (four chains at once keep four sets of per-lane round keys in
        // flight: 128 VGPRs + 560 B of scratch per lane)
        if (SENDER) {
            const blk_t ua = Ua[j], ub = Ub[j];
            ta = aes_encrypt<true>(te, ka, ctr);
            tb = aes_encrypt<true>(te, kb, ctr);
            if (rbit_a) ta = bxor(ta, ua);  // v ^= r_j * u_j (:118-126)
            if (rbit_b) tb = bxor(tb, ub);
        } else {
            const blk_t ch = choices[j];
            ta = aes_encrypt<true>(te, ka, ctr);
            Ua[j] = bxor(bxor(aes_encrypt<true>(te, ka + 44, ctr), ta), ch);  // :212-217
            __builtin_amdgcn_sched_barrier(0);
            tb = aes_encrypt<true>(te, kb, ctr);
            Ub[j] = bxor(bxor(aes_encrypt<true>(te, kb + 44, ctr), tb), ch);
        }
        __builtin_amdgcn_sched_barrier(0);
        // OT i = 128 j + r: bit q of its Block = bit r of row q -- the 128 x 128 bit transpose of the tile, in registers
        uint32_t w[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
        bm_transpose128_wave(w, lane);
        const uint64_t ia = 128 * j + lane, ib = ia + 64;
        const blk_t xa = make_uint4(w[0], w[1], w[2], w[3]), xb = make_uint4(w[4], w[5], w[6], w[7]);
        if (SENDER) {
            const blk_t a0 = cr_hash(te, xa), a1 = cr_hash(te, bxor(xa, delta));
            const blk_t b0 = cr_hash(te, xb), b1 = cr_hash(te, bxor(xb, delta));
            if (ia < count) out[2 * ia] = a0, out[2 * ia + 1] = a1;
            if (ib < count) out[2 * ib] = b0, out[2 * ib + 1] = b1;
        } else {
            const blk_t a0 = cr_hash(te, xa), b0 = cr_hash(te, xb);
            if (ia < count) out[ia] = a0;
            if (ib < count) out[ib] = b0;
        }
        j = column(AZ_TILES + __builtin_amdgcn_readfirstlane(drawn));
    }
}

// ------------------------------------------------------------------------------------------------
// Device-resident OT-extension context = one zappot::ot_ext::Sender or Receiver (ot_ext/mod.rs:28-46, :144-154) for up
// to max_count extensions per call: every device buffer is allocated once, the calls only copy what the caller asks for.
struct sot_alsz_ctx {
    int device;
    uint64_t max_count, max_nj;
    hipStream_t stream;
    uint32_t *d_rks;  // [256][4]: the base-OT Blocks (expanded to round keys by the kernel)
    blk_t *d_ch;      // [max_nj] padded choice Blocks
    blk_t *d_U;       // [128][nj] Blocks (row stride = nj of the last call)
    blk_t *d_out;     // [max_count] (receiver) / [2 max_count] (sender)
    uint8_t *d_bc;    // [128]
    uint32_t *h_rks;  // pinned staging of the base-OT Blocks (+ the 128 base choices behind them)
    uint64_t last_count, last_nj;
    hipEvent_t done;  // recorded behind the last kernel of every call: what a consumer of d_U / d_out on another stream waits for
    std::mutex mu;
};

// live contexts, so that sot_alsz_ctx_send can order itself behind the context whose u-row matrix it is handed
// (a different stream: without the wait the sender's kernel could read U before the receiver's kernel has written it)
static std::mutex g_alsz_reg_mu;
static std::vector<sot_alsz_ctx *> g_alsz_reg;

static void alsz_free(sot_alsz_ctx *a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    void *ptrs[] = {a->d_rks, a->d_ch, a->d_U, a->d_out, a->d_bc};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (a->h_rks) (void)hipHostFree(a->h_rks);
    {
        std::lock_guard<std::mutex> lk(g_alsz_reg_mu);
        for (size_t i = 0; i < g_alsz_reg.size(); i++)
            if (g_alsz_reg[i] == a) {
                g_alsz_reg.erase(g_alsz_reg.begin() + i);
                break;
            }
    }
    if (a->stream) (void)hipStreamSynchronize(a->stream);
    if (a->done) (void)hipEventDestroy(a->done);
    if (a->stream) (void)hipStreamDestroy(a->stream);
    delete a;
}

extern "C" int sot_alsz_ctx_create(int device, uint64_t max_count, sot_alsz_ctx **out) {
    if (!out || max_count == 0) return SOT_EINVAL;
    *out = nullptr;
    SOT_HIP(hipSetDevice(device));
    sot_alsz_ctx *a = new sot_alsz_ctx();
    a->device = device;
    a->max_count = max_count;
    a->max_nj = (max_count + 127) / 128;
    a->last_count = a->last_nj = 0;
    a->stream = nullptr;
    a->done = nullptr;
    a->d_rks = nullptr, a->d_ch = nullptr, a->d_U = nullptr, a->d_out = nullptr, a->d_bc = nullptr, a->h_rks = nullptr;
    const bool ok = hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreateWithFlags(&a->done, hipEventDisableTiming) == hipSuccess &&
                    hipMalloc((void **)&a->d_rks, 256 * 16) == hipSuccess &&
                    hipMalloc((void **)&a->d_ch, a->max_nj * 16) == hipSuccess &&
                    hipMalloc((void **)&a->d_U, (size_t)128 * a->max_nj * 16) == hipSuccess &&
                    hipMalloc((void **)&a->d_out, (size_t)a->max_nj * 128 * 32) == hipSuccess &&
                    hipMalloc((void **)&a->d_bc, 128) == hipSuccess &&
                    hipHostMalloc((void **)&a->h_rks, 256 * 16 + 128, hipHostMallocDefault) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        alsz_free(a);
        sot_set_error("alsz: device allocation failed for %llu OT extensions per call", (unsigned long long)max_count);
        return SOT_ENOMEM;
    }
    {
        std::lock_guard<std::mutex> lk(g_alsz_reg_mu);
        g_alsz_reg.push_back(a);
    }
    *out = a;
    return SOT_OK;
}
extern "C" void sot_alsz_ctx_destroy(sot_alsz_ctx *a) { alsz_free(a); }
// the pointers are fixed at creation; the lock orders the read behind a call in flight on another thread
extern "C" void *sot_alsz_ctx_msgs_dev(sot_alsz_ctx *a) {
    if (!a) return nullptr;
    std::lock_guard<std::mutex> lk(a->mu);
    return a->d_out;
}
extern "C" void *sot_alsz_ctx_u_dev(sot_alsz_ctx *a) {
    if (!a) return nullptr;
    std::lock_guard<std::mutex> lk(a->mu);
    return a->d_U;
}
extern "C" int sot_alsz_ctx_sync(sot_alsz_ctx *a) {
    if (!a) return SOT_EINVAL;
    std::lock_guard<std::mutex> lk(a->mu);
    SOT_HIP(hipSetDevice(a->device));
    SOT_HIP(hipStreamSynchronize(a->stream));
    return SOT_OK;
}

static int alsz_check(const sot_alsz_ctx *a, uint64_t count) {
    if (count == 0 || count % 8) {
        sot_set_error("alsz: the number of OT extensions must be a positive multiple of 8 (ot_ext/mod.rs:58-62)");
        return SOT_EINVAL;
    }
    if (count > a->max_count) {
        sot_set_error("alsz: %llu OT extensions requested from a context created for %llu", (unsigned long long)count,
                      (unsigned long long)a->max_count);
        return SOT_EINVAL;
    }
    return SOT_OK;
}
// copies the nkeys base-OT Blocks into the pinned staging buffer and starts their upload
static int alsz_keys(sot_alsz_ctx *a, const void *base_ots, uint32_t nkeys) {
    SOT_HIP(hipStreamSynchronize(a->stream));  // the staging buffer may still feed the previous call's copy
    memcpy(a->h_rks, base_ots, (size_t)nkeys * 16);
    SOT_HIP(hipMemcpyAsync(a->d_rks, a->h_rks, (size_t)nkeys * 16, hipMemcpyHostToDevice, a->stream));
    return SOT_OK;
}
static inline unsigned alsz_grid(uint64_t nj) {
    uint64_t g = (nj + AZ_TILES - 1) / AZ_TILES;
    return (unsigned)(g > AZ_GRID ? AZ_GRID : g);
}

// Receiver::receive (ot_ext/mod.rs:156-237). choice_bytes: count / 8 bytes, host memory or (choices_on_device != 0)
// device memory. u_rows_out (the message for the sender: [128][count / 8] bytes) and msgs_out ([count] Blocks) are host
// buffers or NULL: with NULL the data stays on the device (sot_alsz_ctx_u_dev: [128][ceil(count / 128)] Blocks,
// sot_alsz_ctx_msgs_dev) and the call returns without waiting for the kernel (sot_alsz_ctx_sync).
extern "C" int sot_alsz_ctx_receive(sot_alsz_ctx *a, const void *base_ots, const uint8_t *choice_bytes, int choices_on_device,
                                    uint64_t count, uint8_t *u_rows_out, void *msgs_out) {
    if (!a || !base_ots || !choice_bytes) return SOT_EINVAL;
    SOT_TRY(alsz_check(a, count));
    std::lock_guard<std::mutex> lk(a->mu);
    SOT_HIP(hipSetDevice(a->device));
    const uint64_t cols = count / 8, nj = (count + 127) / 128;
    SOT_TRY(alsz_keys(a, base_ots, 256));
    if (cols % 16) SOT_HIP(hipMemsetAsync((uint8_t *)a->d_ch + (cols & ~15ull), 0, 16, a->stream));  // pad the last Block
    SOT_HIP(hipMemcpyAsync(a->d_ch, choice_bytes, cols, choices_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                           a->stream));
    hipLaunchKernelGGL(k_alsz<false>, dim3(alsz_grid(nj)), dim3(AZ_THREADS), 0, a->stream, a->d_rks, a->d_ch,
                       make_uint4(0, 0, 0, 0), (const uint8_t *)nullptr, a->d_U, nj, count, a->d_out);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipEventRecord(a->done, a->stream));
    a->last_count = count, a->last_nj = nj;
    if (msgs_out) SOT_HIP(hipMemcpyAsync(msgs_out, a->d_out, count * 16, hipMemcpyDeviceToHost, a->stream));
    // row r of the device matrix is nj Blocks wide, the caller's is count / 8 bytes
    if (u_rows_out) SOT_HIP(hipMemcpy2DAsync(u_rows_out, cols, a->d_U, nj * 16, cols, 128, hipMemcpyDeviceToHost, a->stream));
    if (msgs_out || u_rows_out) SOT_HIP(hipStreamSynchronize(a->stream));
    return SOT_OK;
}

// Sender::send (ot_ext/mod.rs:48-142). The receiver's message comes from host memory (u_rows: [128][count / 8] bytes) or,
// with u_rows == NULL, from device memory (u_dev: [128][ceil(count / 128)] Blocks, e.g. another context's
// sot_alsz_ctx_u_dev on the same GPU). msgs_out: [count][2] Blocks on the host, or NULL (sot_alsz_ctx_msgs_dev).
extern "C" int sot_alsz_ctx_send(sot_alsz_ctx *a, const void *base_ots, const void *base_choices, const uint8_t *u_rows,
                                 const void *u_dev, uint64_t count, void *msgs_out) {
    if (!a || !base_ots || !base_choices || (!u_rows && !u_dev)) return SOT_EINVAL;
    SOT_TRY(alsz_check(a, count));
    std::lock_guard<std::mutex> lk(a->mu);
    SOT_HIP(hipSetDevice(a->device));
    const uint64_t cols = count / 8, nj = (count + 127) / 128;
    SOT_TRY(alsz_keys(a, base_ots, 128));
    blk_t delta;
    memcpy(&delta, base_choices, 16);
    uint8_t *bc = (uint8_t *)(a->h_rks + 256 * 4);
    for (int i = 0; i < 128; i++) bc[i] = (((const uint8_t *)base_choices)[i >> 3] >> (i & 7)) & 1;
    SOT_HIP(hipMemcpyAsync(a->d_bc, bc, 128, hipMemcpyHostToDevice, a->stream));
    const blk_t *U = (const blk_t *)u_dev;
    if (!u_rows) {
        // u_dev is the u-row matrix of a receiver context on this GPU (in-process two-party runs): its kernel runs on that
        // context's stream, so this stream waits for the event recorded behind it. A pointer that belongs to no live
        // context is the caller's own buffer and must be complete before the call (include/flute_sot.h).
        std::lock_guard<std::mutex> rl(g_alsz_reg_mu);
        for (sot_alsz_ctx *p : g_alsz_reg)
            if (p != a && p->d_U == u_dev && p->device == a->device) SOT_HIP(hipStreamWaitEvent(a->stream, p->done, 0));
    }
    if (u_rows) {
        if (cols % 16) SOT_HIP(hipMemsetAsync(a->d_U, 0, (size_t)128 * nj * 16, a->stream));  // rows end inside a Block
        SOT_HIP(hipMemcpy2DAsync(a->d_U, nj * 16, u_rows, cols, cols, 128, hipMemcpyHostToDevice, a->stream));
        U = a->d_U;
    }
    hipLaunchKernelGGL(k_alsz<true>, dim3(alsz_grid(nj)), dim3(AZ_THREADS), 0, a->stream, a->d_rks, (const blk_t *)nullptr,
                       delta, a->d_bc, (blk_t *)U, nj, count, a->d_out);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipEventRecord(a->done, a->stream));
    a->last_count = count, a->last_nj = nj;
    if (msgs_out) {
        SOT_HIP(hipMemcpyAsync(msgs_out, a->d_out, count * 32, hipMemcpyDeviceToHost, a->stream));
        SOT_HIP(hipStreamSynchronize(a->stream));
    }
    return SOT_OK;
}

// one-shot forms: a context for exactly this call
static int alsz_run(bool sender, int device, const void *base_ots, const uint8_t *choice_bytes, const void *base_choices,
                    uint8_t *u_rows, uint64_t count, void *msgs_out) {
    if (!base_ots || !u_rows || !msgs_out || (sender ? !base_choices : !choice_bytes)) return SOT_EINVAL;
    if (count == 0 || count % 8) {
        sot_set_error("alsz: the number of OT extensions must be a positive multiple of 8 (ot_ext/mod.rs:58-62)");
        return SOT_EINVAL;
    }
    sot_alsz_ctx *a = nullptr;
    SOT_TRY(sot_alsz_ctx_create(device, count, &a));
    const int rc = sender ? sot_alsz_ctx_send(a, base_ots, base_choices, u_rows, nullptr, count, msgs_out)
                          : sot_alsz_ctx_receive(a, base_ots, choice_bytes, 0, count, u_rows, msgs_out);
    alsz_free(a);
    return rc;
}

extern "C" int sot_alsz_receive(int device, const void *base_ots, const uint8_t *choice_bytes, uint64_t count,
                                uint8_t *u_rows_out, void *msgs_out) {
    return alsz_run(false, device, base_ots, choice_bytes, nullptr, u_rows_out, count, msgs_out);
}
extern "C" int sot_alsz_send(int device, const void *base_ots, const void *base_choices, const uint8_t *u_rows,
                             uint64_t count, void *msgs_out) {
    return alsz_run(true, device, base_ots, nullptr, base_choices, (uint8_t *)u_rows, count, msgs_out);
}

// ------------------------------------------------------------------------------------------------
// The one entry point that allocates nothing (no counterpart in the reference; for callers that cannot call hipMalloc:
// tight serving loops, harnesses that forbid allocation). Synchronous on the null stream (blocking copies), so NOT usable
// inside a HIP stream capture. Same computation and same kernel as
// sot_alsz_receive, but every device buffer -- the 256 expanded base-OT keys, the padded choice blocks, the u-row
// matrix and the hashed messages -- lives in ONE file-scope device array of a fixed, modest size, so the call makes no
// hipMalloc / hipFree and no host heap allocation either. Each GPU holds its own copy of the array until the process
// ends; its device address is looked up once per device with hipGetSymbolAddress and only that pointer is ever handed
// to kernels and copies. Calls on one GPU share the array: within this process they are serialised by a lock, and the
// array keeps the previous call's contents (the one region that must start at zero, the choice padding, is cleared).
#define AZ_NA_MAX_COUNT (1ull << 16)                 // OT extensions per call
#define AZ_NA_NJ (AZ_NA_MAX_COUNT / 128)             // 128-OT tiles
#define AZ_NA_OFF_RKS 0                              // [256][4] words (the region keeps its 45 056 B)
#define AZ_NA_OFF_CH (AZ_NA_OFF_RKS + 256 * 44 * 4)  // 45056 = 176 * 256: [nj] Blocks
#define AZ_NA_OFF_U (AZ_NA_OFF_CH + AZ_NA_NJ * 16)   // [128][nj] Blocks
#define AZ_NA_OFF_OUT (AZ_NA_OFF_U + 128 * AZ_NA_NJ * 16)
#define AZ_NA_BYTES (AZ_NA_OFF_OUT + AZ_NA_MAX_COUNT * 16)  // 2 150 400 B (2.05 MiB) per GPU
static __device__ __align__(256) unsigned char scratch[AZ_NA_BYTES];

#define AZ_NA_MAX_DEVICES 64
static std::mutex g_na_mu;
static unsigned char *g_na_ptr[AZ_NA_MAX_DEVICES];  // device address of `scratch` on each GPU (nullptr: not looked up yet)

extern "C" uint64_t sot_alsz_noalloc_max_count(void) { return AZ_NA_MAX_COUNT; }

extern "C" int sot_alsz_receive_noalloc(int device, const void *base_ots, const uint8_t *choice_bytes, uint64_t count,
                                        uint8_t *u_rows_out, void *msgs_out) {
    if (!base_ots || !choice_bytes || !u_rows_out || !msgs_out) return SOT_EINVAL;
    if (count == 0 || count % 8) {
        sot_set_error("alsz: the number of OT extensions must be a positive multiple of 8 (ot_ext/mod.rs:58-62)");
        return SOT_EINVAL;
    }
    if (count > AZ_NA_MAX_COUNT) {
        sot_set_error("sot_alsz_receive_noalloc: count %llu exceeds the fixed scratch array (at most %llu OTs per call; "
                      "use sot_alsz_receive for more)",
                      (unsigned long long)count, (unsigned long long)AZ_NA_MAX_COUNT);
        return SOT_EINVAL;
    }
    if (device < 0 || device >= AZ_NA_MAX_DEVICES) {
        sot_set_error("sot_alsz_receive_noalloc: device index %d out of range", device);
        return SOT_EINVAL;
    }
    std::lock_guard<std::mutex> lk(g_na_mu);  // calls that share a GPU's array must not overlap
    SOT_HIP(hipSetDevice(device));
    if (!g_na_ptr[device]) {
        // once per device, with that device current: the array's name is a host-side stand-in, not a device address
        unsigned char *p = nullptr;
        SOT_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        if (!p) {
            sot_set_error("sot_alsz_receive_noalloc: no device address for the scratch array");
            return SOT_EDEVICE;
        }
        g_na_ptr[device] = p;
    }
    unsigned char *const base = g_na_ptr[device];
    uint32_t *const d_rks = (uint32_t *)(base + AZ_NA_OFF_RKS);
    blk_t *const d_ch = (blk_t *)(base + AZ_NA_OFF_CH);
    blk_t *const d_U = (blk_t *)(base + AZ_NA_OFF_U);
    blk_t *const d_out = (blk_t *)(base + AZ_NA_OFF_OUT);
    const uint64_t cols = count / 8, nj = (count + 127) / 128;
    SOT_HIP(hipMemcpy(d_rks, base_ots, 256 * 16, hipMemcpyHostToDevice));  // expanded to round keys by the kernel
    // choices as nj whole Blocks: the bytes past count / 8 must be zero, whatever the last call left there
    SOT_HIP(hipMemset(d_ch, 0, nj * 16));
    SOT_HIP(hipMemcpy(d_ch, choice_bytes, cols, hipMemcpyHostToDevice));
    uint64_t g = (nj + AZ_TILES - 1) / AZ_TILES;
    if (g > AZ_GRID) g = AZ_GRID;
    // U[r][j] (r < 128, j < nj) and out[i] (i < count) are written in full by the kernel before they are read back
    hipLaunchKernelGGL(k_alsz<false>, dim3((unsigned)g), dim3(AZ_THREADS), 0, 0, d_rks, d_ch, make_uint4(0, 0, 0, 0),
                       (const uint8_t *)nullptr, d_U, nj, count, d_out);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipMemcpy(msgs_out, d_out, count * 16, hipMemcpyDeviceToHost));
    // row r of the device matrix is nj Blocks wide, the caller's is count / 8 bytes
    SOT_HIP(hipMemcpy2D(u_rows_out, cols, d_U, nj * 16, cols, 128, hipMemcpyDeviceToHost));
    return SOT_OK;
}
