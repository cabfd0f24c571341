// Microbenchmark: in-place copy of a 512 MiB buffer in 64 KiB tiles gathered from scattered granules, with the structure
// of k_mid (persistent workgroups of 512 threads, two per CU, next tile prefetched into registers). Sweeps the granule
// size and the scatter pattern to price the HBM access pattern of the tile kernels on its own.
//   hipcc --offload-arch=gfx950 -O3 -o gather_copy gather_copy.hip && ./gather_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct pat { uint32_t tmask; uint32_t nslot_bits; uint32_t gshift; };  // gshift: log2(granule / 256 B)

__device__ __forceinline__ uint32_t pdep(uint32_t v, uint32_t mask) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < 24; b++)
        if ((mask >> b) & 1u) { r |= (v & 1u) << b; v >>= 1; }
    return r;
}

// granule index space: 2^21 granules of 256 B (512 MiB). A tile = 256 pieces of 256 B; with gshift g, 2^g consecutive pieces
// form one contiguous granule, so only 8 - g tile bits are scattered (tmask, over the index of 256-B pieces >> g).
template <int STORE>
__global__ void __launch_bounds__(512) k_copy(uint32_t *Z, pat p, uint32_t ntiles) {
    __shared__ uint32_t lds[256 * 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto addr = [&](uint32_t tile, uint32_t slot) -> uint32_t * {
        const uint32_t lo = slot & ((1u << p.gshift) - 1u), hi = slot >> p.gshift;
        const uint32_t g = pdep(hi, p.tmask) | pdep(tile, ~p.tmask & ((1u << (21 - p.gshift)) - 1u));
        return Z + ((((uint64_t)g << p.gshift) | lo) << 6) + lane;
    };
    uint32_t v[32];
    uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = *addr(t, wave + 8 * i);
    for (; t < ntiles; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < 32; i++) lds[(wave + 8 * i) * 64 + lane] = v[i];
        __syncthreads();
        const uint32_t tn = t + gridDim.x < ntiles ? t + gridDim.x : t;
#pragma unroll
        for (int i = 0; i < 32; i++) v[i] = *addr(tn, wave + 8 * i);
        if (STORE) {
#pragma unroll
            for (int i = 0; i < 32; i++) *addr(t, wave + 8 * i) = lds[((wave + 8 * i) ^ 1) * 64 + lane] + 1u;
        }
        __syncthreads();
    }
    if (!STORE && v[0] == 0x12345678u) Z[0] = v[1];
}

// 64-B pieces (4 lanes x 16 B) at a stride of 512 B, the eight pieces of a 512-B word read by eight different workgroups:
// the access pattern of k_conv_unit / k_conv_tile (16-row tiles). Reads all 512 MiB once; for calibrating FETCH_SIZE.
// MAP 0: piece = blockIdx % 8 (so XCD k reads piece k of every word); MAP 1: the eight pieces of a chunk of words on eight
// workgroups of ONE XCD (blockIdx = xcd + 8 * (piece + 8 * s)). WR: also write the piece back.
template <int MAP, int WR>
__global__ void __launch_bounds__(1024) k_piece64(uint32_t *Z, uint32_t *out) {
    const uint32_t quad = threadIdx.x & 3;
    const uint32_t b = blockIdx.x;
    const uint32_t tq = MAP ? (b >> 3) & 7 : b & 7;
    const uint32_t chunk = MAP ? ((b >> 6) << 3) | (b & 7) : b >> 3;
    const uint32_t word = chunk * 256 + (threadIdx.x >> 2);
    uint4 *p = (uint4 *)(Z + (uint64_t)word * 128 + tq * 16 + quad * 4);
    uint4 v = *p;
    if (WR) {
        v.x += 1;
        *p = v;
    } else if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v.x;
}

// 128-B pieces (8 lanes x 16 B) at a stride of 512 B, the four pieces of a word on four workgroups (k_conv_taylor with 32-row
// tiles); for calibrating FETCH_SIZE
__global__ void __launch_bounds__(1024) k_piece128(const uint32_t *Z, uint32_t *out) {
    const uint32_t oct = threadIdx.x & 7, tq = blockIdx.x & 3;
    const uint32_t word = (blockIdx.x >> 2) * 128 + (threadIdx.x >> 3);
    const uint4 v = *(const uint4 *)(Z + (uint64_t)word * 128 + tq * 32 + oct * 4);
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[0] = v.x;
}

int main() {
    uint32_t *Z;
    const size_t bytes = 512ull << 20;
    CK(hipMalloc(&Z, bytes));
    CK(hipMemset(Z, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct named { const char *name; pat p; };
    // bit positions over the index of 256-B pieces: piece = word * 2 + row half (word = 512 B)
    auto wbits = [](std::initializer_list<int> wb, int gshift) { uint32_t m = 0; for (int b : wb) m |= 1u << (b + 1 - gshift); return m; };
    std::vector<named> pats = {
        {"contiguous 64 KiB tiles           ", {0xFFu, 8, 0}},
        {"k_mid:  256 B x 256 (wb 8-12,16,18,19)", {wbits({8, 9, 10, 11, 12, 16, 18, 19}, 0), 8, 0}},
        {"k_mid-like 512 B x 128 (wb 9-12,16,18,19)", {wbits({9, 10, 11, 12, 16, 18, 19}, 1), 7, 1}},
        {"1 KiB x 64 (wb 10-12,16,18,19)     ", {wbits({10, 11, 12, 16, 18, 19}, 2), 6, 2}},
        {"2 KiB x 32 (wb 11,12,16,18,19)     ", {wbits({11, 12, 16, 18, 19}, 3), 5, 3}},
        {"4 KiB x 16 (wb 12,16,18,19)        ", {wbits({12, 16, 18, 19}, 4), 4, 4}},
        {"k_top:  256 B x 256 (wb 12-19)     ", {wbits({12, 13, 14, 15, 16, 17, 18, 19}, 0), 8, 0}},
        {"low:    256 B x 512-like (wb 0-4,12,16,18)", {wbits({0, 1, 2, 3, 4, 12, 16, 18}, 0), 8, 0}},
        {"256 B x 256, stride 128 KiB (wb 8-15)", {wbits({8, 9, 10, 11, 12, 13, 14, 15}, 0), 8, 0}},
        {"256 B x 256, stride 1 KiB (wb 1-8)  ", {wbits({1, 2, 3, 4, 5, 6, 7, 8}, 0), 8, 0}},
    };
    const uint32_t ntiles = 8192;
    for (int store = 1; store >= 0; store--)
        for (auto &n : pats) {
            for (unsigned grid : {512u, 1024u}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; rep++) {
                    CK(hipEventRecord(e0));
                    if (store) hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(512), 0, 0, Z, n.p, ntiles);
                    else hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(512), 0, 0, Z, n.p, ntiles);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                printf("%s %s grid %4u  %.3f ms  %.2f TB/s\n", store ? "read+write" : "read only ", n.name, grid, best,
                       (store ? 2.0 : 1.0) * bytes / best * 1e-9);
            }
        }
    for (int var = 0; var < 4; var++)
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            const dim3 g(8 * 4096), t(1024);
            switch (var) {
                case 0: hipLaunchKernelGGL((k_piece64<0, 0>), g, t, 0, 0, Z, Z); break;
                case 1: hipLaunchKernelGGL((k_piece64<1, 0>), g, t, 0, 0, Z, Z); break;
                case 2: hipLaunchKernelGGL((k_piece64<0, 1>), g, t, 0, 0, Z, Z); break;
                default: hipLaunchKernelGGL((k_piece64<1, 1>), g, t, 0, 0, Z, Z); break;
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s 64-B pieces at stride 512 B, pieces of a word on %s: %.3f ms  %.2f TB/s\n", var & 2 ? "read+write" : "read only ",
                   var & 1 ? "one XCD   " : "eight XCDs", ms, (var & 2 ? 2.0 : 1.0) * bytes / ms * 1e-9);
        }
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_piece128, dim3(4 * 8192), dim3(1024), 0, 0, Z, Z);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("read only  128-B pieces at stride 512 B, pieces of a word on four XCDs: %.3f ms  %.2f TB/s\n", ms, bytes / ms * 1e-9);
    }
    CK(hipGetLastError());
    return 0;
}
