// VALU issue-rate microbenchmark (gfx950): wave-instructions per second for a few integer ops, 16 waves per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define N_IT 8000
template <int OP>
__global__ void __launch_bounds__(1024) k(uint32_t *out, uint32_t seed, uint32_t smask) {
    uint32_t a[16];
    for (int i = 0; i < 16; i++) a[i] = threadIdx.x * 2654435761u + i + seed;
    uint32_t b = seed ^ threadIdx.x, c = seed * 3 + threadIdx.x;
    for (int it = 0; it < N_IT; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (OP == 0) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, c, 0x78);           // vgpr mask
            if (OP == 1) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, smask + i, 0x78);   // sgpr mask
            if (OP == 2) a[i] = a[i] ^ b;
            if (OP == 3) a[i] = __builtin_amdgcn_perm(a[i], b, 0x0c0c0500u);
            if (OP == 4) a[i] = __builtin_amdgcn_alignbit(a[i], a[i], 24);
            if (OP == 5) a[i] = (a[i] & smask) | b;  // v_and_or_b32
            if (OP == 6) a[i] = a[i] + b;
            // a ^ (b & m) with the SGPR mask as src0 / src1 (truth tables for operand orders (m,b,a) and (b,m,a))
            if (OP == 7) a[i] = __builtin_amdgcn_bitop3_b32(smask + i, b, a[i], 0x6a);
            if (OP == 8) a[i] = __builtin_amdgcn_bitop3_b32(b, smask + i, a[i], 0x6a);
            if (OP == 9) a[i] = a[i] ^ (smask + i);             // VOP2 xor with SGPR
            if (OP == 10) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b, 0x0f0f0f0fu, 0x78);  // literal
            if (OP == 11) a[i] = __builtin_amdgcn_bitop3_b32(a[i], a[i], c, 0x78);  // two distinct VGPRs only
        }
        b += c;
    }
    uint32_t r = 0;
    for (int i = 0; i < 16; i++) r ^= a[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int OP>
void run(const char *name, uint32_t *d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int blocks = 256 * 4;
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 1u, 0x0f0f0f0fu);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(1024), 0, 0, d, 2u, 0x0f0f0f0fu);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 16 * N_IT * 17;  // 16 ops + 1 add per iteration
    printf("%-28s %8.3f ms  %.3e wave-instr/s  (%.2f cycles/instr/SIMD at 2.4 GHz)\n", name, ms, winstr / (ms * 1e-3),
           1024 * 2.4e9 / (winstr / (ms * 1e-3)));
}
int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 4 * 1024 * 4);
    run<0>("bitop3 vgpr,vgpr,vgpr", d);
    run<1>("bitop3 vgpr,vgpr,sgpr", d);
    run<2>("v_xor", d);
    run<3>("v_perm", d);
    run<4>("v_alignbit", d);
    run<5>("v_and_or", d);
    run<6>("v_add_u32", d);
    run<7>("bitop3 sgpr,vgpr,vgpr", d);
    run<8>("bitop3 vgpr,sgpr,vgpr", d);
    run<9>("v_xor vgpr,sgpr", d);
    run<10>("bitop3 vgpr,vgpr,literal", d);
    run<11>("bitop3 2 distinct vgprs", d);
    run<0>("bitop3 vgpr,vgpr,vgpr again", d);
    run<1>("bitop3 vgpr,vgpr,sgpr again", d);
    return 0;
}
