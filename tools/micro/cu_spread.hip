// Do all CUs run the same LDS-bound loop at the same speed? One 512-thread workgroup pair per CU (grid = 512, all resident),
// every workgroup records where it ran (XCC_ID, HW_ID) and when it finished (wall_clock64, 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../flute_amd/csrc cu_spread.hip -o cu_spread && ./cu_spread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "aes_dev.h"

__global__ void __launch_bounds__(512) k(blk_t *out, unsigned long long *info, int iters) {
    __shared__ uint32_t te[1][AES_LDS_WORDS];
    aes_lds_fill(te);
    const unsigned long long t0 = wall_clock64();
    blk_t x[4];
#pragma unroll
    for (int c = 0; c < 4; c++) x[c] = make_uint4(threadIdx.x, blockIdx.x, c, 0x1234567);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int c = 0; c < 4; c++) x[c] = bxor(aes_encrypt(te, c_rk_fixed[c & 1], x[c]), x[c]);
    }
    out[(size_t)blockIdx.x * 512 + threadIdx.x] = bxor(bxor(x[0], x[1]), bxor(x[2], x[3]));
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long t1 = wall_clock64();
        const uint32_t xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));   // XCC_ID[3:0]
        const uint32_t hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));    // HW_ID
        info[3 * blockIdx.x + 0] = t0;
        info[3 * blockIdx.x + 1] = t1;
        info[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
}

int main() {
    const int grid = 512, iters = 800;
    blk_t *out;
    unsigned long long *info;
    hipMalloc(&out, (size_t)grid * 512 * 16);
    hipMalloc(&info, grid * 24);
    for (int i = 0; i < 20; i++) hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, info, iters);  // warm-up
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), 0, 0, out, info, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(3 * grid);
    hipMemcpy(h.data(), info, grid * 24, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < grid; b++) tmin = std::min(tmin, h[3 * b]);
    double sum[16] = {0}, mx[16] = {0}, mn[16];
    int cnt[16] = {0};
    for (int x = 0; x < 16; x++) mn[x] = 1e30;
    std::vector<double> all;
    for (int b = 0; b < grid; b++) {
        const int xcc = (int)(h[3 * b + 2] >> 32) & 15;
        const double dur = (h[3 * b + 1] - h[3 * b]) * 0.01;  // us
        sum[xcc] += dur, cnt[xcc]++, mx[xcc] = std::max(mx[xcc], dur), mn[xcc] = std::min(mn[xcc], dur);
        all.push_back(dur);
    }
    for (int x = 0; x < 16; x++)
        if (cnt[x]) printf("XCC %2d: %3d workgroups, run time min %8.1f mean %8.1f max %8.1f us\n", x, cnt[x], mn[x], sum[x] / cnt[x], mx[x]);
    std::sort(all.begin(), all.end());
    printf("all: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us\n", all[0], all[grid / 10], all[grid / 2], all[grid * 9 / 10], all[grid - 1]);
    // workgroups per (xcc, cu): HW_ID bits: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    int hist[8] = {0};
    std::vector<int> per(16 * 4096, 0);
    for (int b = 0; b < grid; b++) {
        const uint32_t hw = (uint32_t)h[3 * b + 2];
        const int xcc = (int)(h[3 * b + 2] >> 32) & 15;
        per[xcc * 4096 + ((hw >> 8) & 0xFF)]++;
    }
    int ncu = 0;
    for (int v : per)
        if (v) hist[v < 7 ? v : 7]++, ncu++;
    // start / end of the workgroups that shared a CU, for a few CUs
    int shown = 0;
    for (int key = 0; key < 16 * 4096 && shown < 6; key++) {
        if (!per[key]) continue;
        printf("XCC %d SE/SH/CU 0x%02x:", key / 4096, key % 4096);
        for (int b = 0; b < grid; b++) {
            const uint32_t hw = (uint32_t)h[3 * b + 2];
            const int xcc = (int)(h[3 * b + 2] >> 32) & 15;
            if (xcc * 4096 + (int)((hw >> 8) & 0xFF) == key)
                printf("  wg %3d [%7.1f .. %7.1f]", b, (h[3 * b] - tmin) * 0.01, (h[3 * b + 1] - tmin) * 0.01);
        }
        printf("\n");
        shown++;
    }
    printf("distinct (XCC, SE/SH/CU) ids seen: %d; ids with 1 / 2 / 3 / 4 workgroups: %d / %d / %d / %d\n", ncu, hist[1], hist[2], hist[3], hist[4]);
    return 0;
}
