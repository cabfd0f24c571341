// AES T-table rate on one MI355X without any global-memory traffic: what fraction of the LDS-lookup ceiling
// (256 CUs x 2.4 GHz / 2 cycles per wave64 ds_read_b32 x 64 lanes / 160 lookups = 1.2288e11 blocks/s) does the AES of
// csrc/aes_dev.h reach by itself, by chains per thread and waves per CU -- and what does the bare lookup stream reach.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../flute_amd/csrc aes_rate.hip -o aes_rate && ./aes_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "aes_dev.h"

#define CK(x)                                                            \
    do {                                                                 \
        hipError_t e_ = (x);                                             \
        if (e_ != hipSuccess) {                                          \
            printf("%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                     \
        }                                                                \
    } while (0)

// CH independent chains per thread, ITERS dependent encryptions each: x = AES_k(x) ^ x
template <int CH, int THREADS, int WPE = 4>
__global__ void __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k_aes(blk_t *out, int iters, int stagger) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    // break the lockstep of the waves that share a SIMD (they all leave the barrier in the same cycle): wave w idles
    // (w / 4) * stagger * 64 cycles (waves w, w + 4, ... share a SIMD)
    {
        // mode (stagger >> 8): 0 = by SIMD-mate group (wave / 4), 1 = by wave index in the workgroup, 2 = by the hardware's
        // wave slot and SIMD (HW_ID: unique per CU whatever workgroups share it)
        const int mode = stagger >> 8, st = stagger & 255;
        int slot = threadIdx.x >> 8;
        if (mode == 1) slot = threadIdx.x >> 6;
        if (mode == 2) {
            const uint32_t hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (5 << 11));  // wave_id[3:0], simd_id[5:4]
            slot = (hw & 15u) * 4 + ((hw >> 4) & 3u);
        }
        for (int i = 0; i < slot * st; i++) __builtin_amdgcn_s_sleep(1);
    }
    blk_t x[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) x[c] = make_uint4(threadIdx.x, blockIdx.x, c, 0x1234567);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = bxor(aes_encrypt(te, c_rk_fixed[c & 1 ? AES_K1 : AES_K0], x[c]), x[c]);
    }
    blk_t s = x[0];
#pragma unroll
    for (int c = 1; c < CH; c++) s = bxor(s, x[c]);
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = s;
}

// the bare lookup stream: 16 independent ds_read_b32 per "round" whose addresses come from the previous round's values
// through one v_perm each (the AES address pattern without its XORs): 160 lookups per "block"
template <int CH, int THREADS>
__global__ void __launch_bounds__(THREADS) k_lookups(blk_t *out, int iters) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    const char *tb = (const char *)te[0];
    const uint32_t loff = (threadIdx.x & 31) << 2;
    uint32_t s[CH][4];
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) s[c][j] = threadIdx.x * 0x9E3779B9u + blockIdx.x + c * 77 + j;
    for (int it = 0; it < iters * 10; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            uint32_t t[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t a = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t ad = __builtin_amdgcn_perm(s[c][(j + k) & 3], loff, 0x0c0c0000u | ((4u + k) << 8));
                    a += *(const uint32_t *)(tb + ad + (k & 1 ? 128 : 0));
                }
                t[j] = a;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) s[c][j] = t[j];
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < CH; c++) r ^= s[c][0] ^ s[c][1] ^ s[c][2] ^ s[c][3];
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = make_uint4(r, 0, 0, 0);
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

template <int CH, int THREADS, bool LOOK, int WPE = 4>
static void run(blk_t *out, int wg_per_cu, int iters, int stagger = 0) {
    const int grid = 256 * wg_per_cu;
    const double ms = time_ms([&] {
        if (LOOK)
            hipLaunchKernelGGL((k_lookups<CH, THREADS>), dim3(grid), dim3(THREADS), 0, 0, out, iters);
        else
            hipLaunchKernelGGL((k_aes<CH, THREADS, WPE>), dim3(grid), dim3(THREADS), 0, 0, out, iters, stagger);
    });
    const double blocks = (double)grid * THREADS * CH * iters;
    printf("stagger %2d %-8s chains %d  threads %4d  wg/CU %d (%2d waves/CU): %7.3f ms  %.3e blocks/s  %.3f of the lookup ceiling\n",
           stagger, LOOK ? "lookups" : "aes", CH, THREADS, wg_per_cu, wg_per_cu * THREADS / 64, ms, blocks / (ms * 1e-3),
           blocks / (ms * 1e-3) / 1.2288e11);
}

int main() {
    blk_t *out;
    CK(hipMalloc(&out, (size_t)8 * 1024 * 1024 * 16));
    const int it = 400;
    // the clocks ramp up over the first tens of milliseconds of load: without this warm-up every sweep below shows a trend
    // along its own order (which read as "staggered waves are faster" until it was run in random order)
    for (int i = 0; i < 30; i++) hipLaunchKernelGGL((k_aes<4, 512, 4>), dim3(512), dim3(512), 0, 0, out, 400, 0);  // ~130 ms of load
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        run<4, 512, false, 4>(out, 15, 2, 0);
        run<4, 512, false, 4>(out, 15, 4, 0);
        run<4, 512, false, 4>(out, 15, 8, 0);
        run<4, 512, false, 4>(out, 15, 16, 0);
        run<4, 512, false, 4>(out, 8, 50, 0);
        run<4, 1024, false, 4>(out, 8, 4, 0);
        run<4, 256, false, 4>(out, 30, 4, 0);
    }
    return 0;
}
