// Microbenchmark for the intra-word phase of k_fft_low<1>: acc[k] ^= v[j] & M[j][k] with the 16 x 32 masks of a level
// broadcast from LDS (ds_read_b128, same address in every lane).
//   A: 16 waves per CU, one row half per wave (today's kernel): one broadcast read per 4 XANDs
//   B:  8 waves per CU, both row halves in one wave: one broadcast read per 8 XANDs, 2 waves per SIMD
//   C: 16 waves per CU, both row halves in one wave (twice the data per CU: the other half of the tile lives in
//      registers instead of LDS): one broadcast read per 8 XANDs, 4 waves per SIMD
// Same number of XANDs in both; prints ms and XAND wave-instructions/s.
//   hipcc --offload-arch=gfx950 -O3 -o mask_share mask_share.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define XAND(a, b, c) __builtin_amdgcn_bitop3_b32((a), (b), (c), 0x78)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int HALVES, int THREADS>
__global__ void __launch_bounds__(THREADS) k(uint32_t *out, const uint32_t *seed, int reps) {
    __shared__ uint4 M[16 * 128];  // 16 waves x one 2 KiB mask block
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t i = threadIdx.x; i < 16 * 128; i += blockDim.x) M[i] = make_uint4(seed[i & 255], seed[(i + 1) & 255], seed[(i + 2) & 255], seed[(i + 3) & 255]);
    __syncthreads();
    uint32_t acc[HALVES][32], v[HALVES][16];
#pragma unroll
    for (int h = 0; h < HALVES; h++) {
#pragma unroll
        for (int k = 0; k < 32; k++) acc[h][k] = seed[(k + h) & 255] + lane;
#pragma unroll
        for (int j = 0; j < 16; j++) v[h][j] = seed[(j + 7 * h) & 255] ^ lane;
    }
    for (int r = 0; r < reps; r++) {
        const uint4 *m = M + ((wave + r) & 15) * 128;  // a different block every round: the reads cannot be hoisted
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
            for (int kq = 0; kq < 8; kq++) {
                const uint4 t = m[j * 8 + kq];
#pragma unroll
                for (int h = 0; h < HALVES; h++) {
                    acc[h][4 * kq + 0] = XAND(acc[h][4 * kq + 0], v[h][j], t.x);
                    acc[h][4 * kq + 1] = XAND(acc[h][4 * kq + 1], v[h][j], t.y);
                    acc[h][4 * kq + 2] = XAND(acc[h][4 * kq + 2], v[h][j], t.z);
                    acc[h][4 * kq + 3] = XAND(acc[h][4 * kq + 3], v[h][j], t.w);
                }
            }
        }
        // keep the loop from being collapsed: feed the accumulators back into the operands
#pragma unroll
        for (int h = 0; h < HALVES; h++)
#pragma unroll
            for (int j = 0; j < 16; j++) v[h][j] ^= acc[h][j] >> 1;
    }
    uint32_t s = 0;
#pragma unroll
    for (int h = 0; h < HALVES; h++)
#pragma unroll
        for (int k = 0; k < 32; k++) s ^= acc[h][k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    uint32_t *out, *seed;
    CK(hipMalloc(&out, 1024 * 1024 * 4));
    CK(hipMalloc(&seed, 1024));
    uint32_t h[256];
    for (int i = 0; i < 256; i++) h[i] = 0x9e3779b9u * (i + 1);
    CK(hipMemcpy(seed, h, 1024, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 400;
    for (int var = 0; var < 3; var++)
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL((k<1, 1024>), dim3(256), dim3(1024), 0, 0, out, seed, reps);
            else if (var == 1) hipLaunchKernelGGL((k<2, 512>), dim3(256), dim3(512), 0, 0, out, seed, reps);
            else hipLaunchKernelGGL((k<2, 1024>), dim3(256), dim3(1024), 0, 0, out, seed, reps);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double xands = 256.0 * (var == 2 ? 32 : 16) * 512 * reps;  // wave-instructions: 256 CUs x wave-halves x 512 per level
            printf("%s: %.3f ms, %.3e XAND wave-instr/s, %.2f cycles per XAND and SIMD at 2.4 GHz\n",
                   var == 2 ? "C both row halves per wave (16 waves/CU)" : var ? "B both row halves per wave ( 8 waves/CU)" : "A one row half per wave    (16 waves/CU)", ms,
                   xands / (ms * 1e-3), ms * 1e-3 * 2.4e9 / (xands / 1024));
        }
    CK(hipGetLastError());
    return 0;
}
