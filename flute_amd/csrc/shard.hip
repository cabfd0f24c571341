// ONE silent-OT expansion sharded over the GPUs of a node (BASELINE config 3): tree sets and bit rows.
//
// The reference's three loci of parallelism are the tree groups of the PPRF (pprf.rs:128, `step_by(8 * num_threads)`),
// the 128 bit rows of the QuasiCyclic products (quasi_cyclic_encode.rs:35-46) and the column tiles of copy_out
// (:140-151). With `world` ranks (one process per GPU, 2, 4 or 8), rank r owns
//   GGM     tree sets [set0_r, set0_r + nl_r): its TreeGrps and the leaf tiles `set + nsets * k` of V, both roles
//           (or, with the GGM stage replicated, every rank expands all sets and there is no exchange #1);
//   QC      bit rows [r * rpr, (r + 1) * rpr), rpr = 128 / world, of EVERY leaf Block, both roles;
//   hash    outputs k in [r * chunk, r * chunk + n_local).
// exchange #1 (all-to-all): a rank sends, to every peer q, bit rows R_q of its own leaf Blocks -- 16 / world bytes per
//   Block and role; what it keeps is its own slice. On arrival the slices of both roles are combined into "slot Blocks"
//   W[leaf] = sender rows R_r | receiver rows R_r << rpr, so that one pass of the (lane = row) transform kernels serves
//   2 * rpr rows: both directions multiply the same public polynomial a, all twiddle / multiplier masks are shared.
// exchange #2 (all-to-all): the encoded slot Blocks of outputs [q * chunk, (q + 1) * chunk) go to rank q -- only their
//   2 * rpr live bits, packed (32 / world bytes per output: k_qc_finish stores them that way) --, which re-assembles
//   whole Blocks B[k], A[k] from the `world` slices and runs the hashes / MT bits on its n_local outputs.
// With 4 or more ranks a rank holds at most 64 live bit rows: the transform kernels then skip the upper row half of every
//   tile and the unused 16-row slices of the conversion passes (qc_set_rows_active).
// The library moves no bytes between GPUs: the host (flute_amd/dist.py over torch.distributed = RCCL, or the Rust shim
// over RCCL's ncclSend/ncclRecv) runs the two all-to-alls on the device buffers these entry points fill and consume.
// Results are bit-identical to the one-GPU context for the same inputs (tests/test_gpu_shard.py).
#include "sot_internal.h"

template <int LW>  // LW = log2(world)
struct slice_ops;
template <>
struct slice_ops<1> {  // 64 rows per rank
    typedef uint2 T;
    static __device__ __forceinline__ T get(const blk_t &v, uint32_t q) { return q ? make_uint2(v.z, v.w) : make_uint2(v.x, v.y); }
    static __device__ __forceinline__ blk_t slots(T s, T r) { return make_uint4(s.x, s.y, r.x, r.y); }
    static __device__ __forceinline__ void put(blk_t &B, blk_t &A, uint32_t r, const blk_t &e) {
        if (r) {
            B.z = e.x, B.w = e.y, A.z = e.z, A.w = e.w;
        } else {
            B.x = e.x, B.y = e.y, A.x = e.z, A.y = e.w;
        }
    }
};
template <>
struct slice_ops<2> {  // 32 rows per rank
    typedef uint32_t T;
    // (selection by shifts of the two 64-bit halves, not a 4-way ternary over the fields: for a run-time q hipcc 7.2
    // turned the ternary chain into a load from a selected ADDRESS and left that address undefined on one path)
    static __device__ __forceinline__ T get(const blk_t &v, uint32_t q) {
        const uint64_t h = (q & 2u) ? (((uint64_t)v.w << 32) | v.z) : (((uint64_t)v.y << 32) | v.x);
        return (uint32_t)(h >> (32 * (q & 1u)));
    }
    static __device__ __forceinline__ blk_t slots(T s, T r) { return make_uint4(s, r, 0u, 0u); }
    static __device__ __forceinline__ void put(blk_t &B, blk_t &A, uint32_t r, const blk_t &e) {
        switch (r) {
            case 0: B.x = e.x, A.x = e.y; break;
            case 1: B.y = e.x, A.y = e.y; break;
            case 2: B.z = e.x, A.z = e.y; break;
            default: B.w = e.x, A.w = e.y; break;
        }
    }
};
template <>
struct slice_ops<3> {  // 16 rows per rank
    typedef uint16_t T;
    static __device__ __forceinline__ T get(const blk_t &v, uint32_t q) {
        const uint64_t h = (q & 4u) ? (((uint64_t)v.w << 32) | v.z) : (((uint64_t)v.y << 32) | v.x);
        return (uint16_t)(h >> (16 * (q & 3u)));
    }
    static __device__ __forceinline__ blk_t slots(T s, T r) { return make_uint4((uint32_t)s | ((uint32_t)r << 16), 0u, 0u, 0u); }
    static __device__ __forceinline__ void put(blk_t &B, blk_t &A, uint32_t r, const blk_t &e) {
        const uint32_t sh = 16 * (r & 1), b = (e.x & 0xFFFFu) << sh, a = (e.x >> 16) << sh;
        switch (r >> 1) {
            case 0: B.x |= b, A.x |= a; break;
            case 1: B.y |= b, A.y |= a; break;
            case 2: B.z |= b, A.z |= a; break;
            default: B.w |= b, A.w |= a; break;
        }
    }
};

struct shard_geom {
    uint32_t rank, world, nsets, nl_max;
    uint64_t nk, cols;
};
__device__ __forceinline__ void sets_of(uint32_t nsets, uint32_t world, uint32_t r, uint32_t &set0, uint32_t &nl) {
    const uint32_t base = nsets / world, rem = nsets % world;
    set0 = r * base + (r < rem ? r : rem);
    nl = base + (r < rem ? 1u : 0u);
}

// exchange #1, sending side. Message for peer q: [role(2)][k < nk][sl < nl_max][j < 128] slices; entry (k, sl, j) is
// the leaf Block V[128 c + j] of tile c = set0 + sl + nsets k. Entries with sl >= nl or c >= cols are not written
// (the receiving side skips the same entries).
// roles: bit 0 = the sender's leaves V0, bit 1 = the receiver's V1 (the two halves of a message can be packed and sent
// separately: the sender's slices travel while the receiver's trees still expand)
template <int LW>
__global__ void __launch_bounds__(256)
    k_pack_leaves(const blk_t *__restrict__ V0, const blk_t *__restrict__ V1, shard_geom g, unsigned char *__restrict__ send,
                  uint64_t x1_bytes, uint32_t roles) {
    typedef typename slice_ops<LW>::T T;
    uint32_t set0, nl;
    sets_of(g.nsets, g.world, g.rank, set0, nl);
    const uint64_t items = g.nk * nl * 128, rs = g.nk * g.nl_max * 128;  // rs: entries per role
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(it & 127);
        const uint64_t t = it >> 7;
        const uint32_t sl = (uint32_t)(t % nl);
        const uint64_t k = t / nl;
        const uint64_t c = set0 + sl + (uint64_t)g.nsets * k;
        if (c >= g.cols) continue;
        const uint64_t lc = (k * g.nl_max + sl) * 128 + j;
        if (roles & 1u) {
            const blk_t v0 = V0[128 * c + j];
#pragma unroll
            for (uint32_t q = 0; q < (1u << LW); q++) ((T *)(send + q * x1_bytes))[lc] = slice_ops<LW>::get(v0, q);
        }
        if (roles & 2u) {
            const blk_t v1 = V1[128 * c + j];
#pragma unroll
            for (uint32_t q = 0; q < (1u << LW); q++) ((T *)(send + q * x1_bytes))[rs + lc] = slice_ops<LW>::get(v1, q);
        }
    }
}

// exchange #1, receiving side: recv = [source rank r][the message above]; W[leaf] = slot Block
// slot Blocks are stored with their 2 * rpr live bits only: 16 / 8 / 4 bytes per leaf at 2 / 4 / 8 ranks (ld_leaf, qc.hip)
template <int LW>
__device__ __forceinline__ void st_slot(blk_t *W, uint64_t idx, blk_t v) {
    if (LW == 1)
        W[idx] = v;
    else if (LW == 2)
        ((uint2 *)W)[idx] = make_uint2(v.x, v.y);
    else
        ((uint32_t *)W)[idx] = v.x;
}
template <int LW>
__global__ void __launch_bounds__(256)
    k_unpack_leaves(const unsigned char *__restrict__ recv, shard_geom g, uint64_t x1_bytes, blk_t *__restrict__ W) {
    typedef typename slice_ops<LW>::T T;
    const uint64_t rs = g.nk * g.nl_max * 128;
    const uint64_t items = rs << LW;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(it / rs);
        const uint64_t lc = it - (uint64_t)r * rs;
        const uint32_t j = (uint32_t)(lc & 127);
        const uint64_t t = lc >> 7;
        const uint32_t sl = (uint32_t)(t % g.nl_max);
        const uint64_t k = t / g.nl_max;
        uint32_t set0, nl;
        sets_of(g.nsets, g.world, r, set0, nl);
        const uint64_t c = set0 + sl + (uint64_t)g.nsets * k;
        if (sl >= nl || c >= g.cols) continue;
        const T *src = (const T *)(recv + r * x1_bytes);
        st_slot<LW>(W, 128 * c + j, slice_ops<LW>::slots(src[lc], src[rs + lc]));
    }
}

// replicated GGM stage: every rank holds all leaves of both roles and cuts its own rows out (no exchange #1)
template <int LW>
__global__ void __launch_bounds__(256)
    k_slots_local(const blk_t *__restrict__ V0, const blk_t *__restrict__ V1, uint32_t rank, uint64_t nV, blk_t *__restrict__ W) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nV; i += (uint64_t)gridDim.x * blockDim.x)
        st_slot<LW>(W, i, slice_ops<LW>::slots(slice_ops<LW>::get(V0[i], rank), slice_ops<LW>::get(V1[i], rank)));
}

// exchange #2, receiving side: recv = [source rank r][kk < chunk] slot Blocks of outputs k0 + kk (rows R_r of B in slots
// [0, rpr), of A in [rpr, 2 rpr)) -> whole Blocks
template <int LW>
__global__ void __launch_bounds__(256)
    k_unpack_enc(const blk_t *__restrict__ recv, uint64_t chunk, uint64_t n_local, blk_t *__restrict__ B, blk_t *__restrict__ A) {
    for (uint64_t kk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; kk < n_local; kk += (uint64_t)gridDim.x * blockDim.x) {
        blk_t b = make_uint4(0, 0, 0, 0), a = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t r = 0; r < (1u << LW); r++) {
            // rank r's message: the 2 * rpr live bits of its slot Blocks, packed (16, 8 or 4 bytes per output)
            blk_t slot = make_uint4(0, 0, 0, 0);
            if (LW == 1) {
                slot = recv[r * chunk + kk];
            } else if (LW == 2) {
                const uint2 t = ((const uint2 *)recv)[r * chunk + kk];
                slot.x = t.x, slot.y = t.y;
            } else {
                slot.x = ((const uint32_t *)recv)[r * chunk + kk];
            }
            slice_ops<LW>::put(b, a, r, slot);
        }
        B[kk] = b;
        A[kk] = a;
    }
}

static inline unsigned sgrid(uint64_t items) {
    uint64_t g = (items + 255) / 256;
    return (unsigned)(g < 1 ? 1 : g > 16384 ? 16384 : g);
}
static shard_geom geom_of(const sot_ctx *c) {
    shard_geom g;
    g.rank = c->sh.rank, g.world = c->sh.world, g.nsets = (uint32_t)c->nsets, g.nl_max = c->sh.nl_max;
    g.nk = c->sh.nk, g.cols = c->conf.cols;
    return g;
}

int shard_init(sot_ctx *c, uint32_t rank, uint32_t world, int ggm_replicated) {
    const sot_conf &cf = c->conf;
    if ((world != 2 && world != 4 && world != 8) || rank >= world) {
        sot_set_error("sharded context: world must be 2, 4 or 8 and rank < world (got rank %u of %u)", rank, world);
        return SOT_EINVAL;
    }
    if (!ggm_replicated && c->nsets < world) {
        sot_set_error("sharded context: %llu tree sets cannot be dealt to %u ranks", (unsigned long long)c->nsets, world);
        return SOT_EINVAL;
    }
    shard_state &s = c->sh;
    s.rank = rank, s.world = world, s.rpr = 128 / world;
    s.ggm_replicated = ggm_replicated != 0;
    const uint32_t base = (uint32_t)c->nsets / world, rem = (uint32_t)c->nsets % world;
    s.nl_max = base + (rem ? 1u : 0u);
    s.nk = (cf.cols + c->nsets - 1) / c->nsets;
    s.chunk = ((cf.n + world - 1) / world + 63) / 64 * 64;
    s.k0 = (uint64_t)rank * s.chunk;
    s.n_local = s.k0 >= cf.n ? 0 : (cf.n - s.k0 < s.chunk ? cf.n - s.k0 : s.chunk);
    s.x1_bytes = ggm_replicated ? 0 : 2 * s.nk * s.nl_max * 128 * (16 / world);
    s.x2_bytes = s.chunk * (2 * s.rpr / 8);  // the 2 * rpr live bits of every slot Block
    if (!ggm_replicated) shard_sets_of((uint32_t)c->nsets, world, rank, &c->ggm_set0, &c->ggm_nsets);
    c->n_out = s.n_local;
    qc_set_rows_active(c, 2 * s.rpr);  // slot Blocks: sender rows, receiver rows, zeros (world >= 4: at most 64 live rows)
    SOT_HIP(hipSetDevice(c->device));
    SOT_TRY(dev_alloc(c, (void **)&s.W, cf.cols * 128 * 16));
    return SOT_OK;
}

void shard_free(sot_ctx *c) {
    if (c->sh.W) (void)hipFree(c->sh.W);
    c->sh.W = nullptr;
}

#define SHARD_CTX(c)                                                  \
    if (!(c)) {                                                       \
        sot_set_error("null context");                                \
        return SOT_EINVAL;                                            \
    }                                                                 \
    std::lock_guard<std::recursive_mutex> _lk((c)->mu);               \
    if ((c)->sh.world < 2) {                                          \
        sot_set_error("not a sharded context (sot_ctx_create_sharded)"); \
        return SOT_ESTATE;                                            \
    }
#define BY_WORLD(c, call1, call2, call3) \
    switch ((c)->sh.world) {             \
        case 2: call1; break;            \
        case 4: call2; break;            \
        default: call3; break;           \
    }

extern "C" int sot_ctx_create_sharded(uint64_t n, uint64_t scaler, int device, uint32_t rank, uint32_t world,
                                      int ggm_replicated, sot_ctx **out) {
    if (!out) return SOT_EINVAL;
    *out = nullptr;
    if ((world != 2 && world != 4 && world != 8) || rank >= world) {
        sot_set_error("sharded context: world must be 2, 4 or 8 and rank < world (got rank %u of %u)", rank, world);
        return SOT_EINVAL;
    }
    sot_ctx *c = nullptr;
    SOT_TRY(sot_ctx_create(n, scaler, device, &c));
    const int rc = shard_init(c, rank, world, ggm_replicated);
    if (rc != SOT_OK) {
        sot_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return SOT_OK;
}

extern "C" int sot_shard_info_get(const sot_ctx *c, sot_shard_info *o) {
    if (!c || !o) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(const_cast<sot_ctx *>(c)->mu);
    const shard_state &s = c->sh;
    o->rank = s.rank, o->world = s.world;
    o->set0 = c->ggm_set0, o->nsets_local = c->ggm_nsets;
    o->ggm_replicated = s.ggm_replicated ? 1 : 0;
    o->rows_per_rank = s.rpr;
    o->k0 = s.world > 1 ? s.k0 : 0, o->n_local = c->n_out, o->chunk = s.world > 1 ? s.chunk : c->conf.n;
    o->x1_bytes_per_peer = s.x1_bytes, o->x2_bytes_per_peer = s.world > 1 ? s.x2_bytes : 0;
    return SOT_OK;
}

static int need_expanded(sot_ctx *c, uint32_t roles = 3) {
    if (((roles & 1u) && !c->role[0].expanded) || ((roles & 2u) && !c->role[1].expanded)) {
        sot_set_error("sharded context: both the sender and the receiver instance must be expanded first");
        return SOT_ESTATE;
    }
    return SOT_OK;
}

extern "C" int sot_shard_pack_leaves_role(sot_ctx *c, void *send_dev, uint32_t roles);
extern "C" int sot_shard_pack_leaves(sot_ctx *c, void *send_dev) { return sot_shard_pack_leaves_role(c, send_dev, 3u); }
extern "C" int sot_shard_pack_leaves_role(sot_ctx *c, void *send_dev, uint32_t roles) {
    SHARD_CTX(c);
    if (!send_dev || roles < 1 || roles > 3) return SOT_EINVAL;
    if (c->sh.ggm_replicated) {
        sot_set_error("sharded context with a replicated GGM stage has no exchange #1: call sot_shard_local_leaves");
        return SOT_ESTATE;
    }
    SOT_TRY(need_expanded(c, roles));
    SOT_HIP(hipSetDevice(c->device));
    const shard_geom g = geom_of(c);
    const unsigned grid = sgrid(g.nk * c->ggm_nsets * 128);
    unsigned char *s = (unsigned char *)send_dev;
    BY_WORLD(c, SOT_LAUNCH(c, KID_SHARD_PACK, k_pack_leaves<1>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, g, s, c->sh.x1_bytes, roles),
             SOT_LAUNCH(c, KID_SHARD_PACK, k_pack_leaves<2>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, g, s, c->sh.x1_bytes, roles),
             SOT_LAUNCH(c, KID_SHARD_PACK, k_pack_leaves<3>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, g, s, c->sh.x1_bytes, roles));
    SOT_HIP(hipGetLastError());
    SOT_SYNC(c);  // the collective runs on another stream, unless the context is stream-ordered on the collective's stream
    return SOT_OK;
}

extern "C" int sot_shard_unpack_leaves(sot_ctx *c, const void *recv_dev) {
    SHARD_CTX(c);
    if (!recv_dev) return SOT_EINVAL;
    if (c->sh.ggm_replicated) {
        sot_set_error("sharded context with a replicated GGM stage has no exchange #1: call sot_shard_local_leaves");
        return SOT_ESTATE;
    }
    SOT_TRY(need_expanded(c));
    SOT_HIP(hipSetDevice(c->device));
    const shard_geom g = geom_of(c);
    const unsigned grid = sgrid((g.nk * g.nl_max * 128) * c->sh.world);
    const unsigned char *r = (const unsigned char *)recv_dev;
    BY_WORLD(c, SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_leaves<1>, dim3(grid), dim3(256), r, g, c->sh.x1_bytes, c->sh.W),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_leaves<2>, dim3(grid), dim3(256), r, g, c->sh.x1_bytes, c->sh.W),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_leaves<3>, dim3(grid), dim3(256), r, g, c->sh.x1_bytes, c->sh.W));
    SOT_HIP(hipGetLastError());
    SOT_SYNC(c);  // the caller may reuse recv_dev
    c->sh.slots_ready = true;
    c->sh.enc_slots_ready = false;
    return SOT_OK;
}

extern "C" int sot_shard_local_leaves(sot_ctx *c) {
    SHARD_CTX(c);
    if (!c->sh.ggm_replicated) {
        sot_set_error("sharded GGM stage: the leaves of other ranks' tree sets come through exchange #1");
        return SOT_ESTATE;
    }
    SOT_TRY(need_expanded(c));
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nV = c->conf.cols * 128;
    const unsigned grid = sgrid(nV);
    BY_WORLD(c, SOT_LAUNCH(c, KID_SHARD_UNPACK, k_slots_local<1>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, c->sh.rank, nV, c->sh.W),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_slots_local<2>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, c->sh.rank, nV, c->sh.W),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_slots_local<3>, dim3(grid), dim3(256), c->role[0].V, c->role[1].V, c->sh.rank, nV, c->sh.W));
    SOT_HIP(hipGetLastError());
    c->sh.slots_ready = true;
    c->sh.enc_slots_ready = false;
    return SOT_OK;
}

extern "C" int sot_shard_qc_encode(sot_ctx *c, const uint64_t *points, uint64_t npoints, void *send_dev) {
    SHARD_CTX(c);
    if (!send_dev) return SOT_EINVAL;
    if (!c->sh.slots_ready) {
        sot_set_error("sot_shard_qc_encode before the leaf slices are in place (sot_shard_unpack_leaves / _local_leaves)");
        return SOT_ESTATE;
    }
    // the receiver's noise indicator sb replaces its bit row 0, which lives on rank 0: only that rank reads the points
    if (c->sh.rank == 0 && (!points || npoints != c->conf.parts)) {
        sot_set_error("receiver encode needs the %llu noisy points (rank 0 holds the receiver's bit row 0)",
                      (unsigned long long)c->conf.parts);
        return SOT_EINVAL;
    }
    SOT_HIP(hipSetDevice(c->device));
    // outputs past n (the padding of the last ranks' chunks) are sent but never read; keep them defined
    const uint64_t total = (uint64_t)c->sh.world * c->sh.chunk;
    if (total > c->conf.n)
        SOT_HIP(hipMemsetAsync((uint8_t *)send_dev + c->conf.n * (2 * c->sh.rpr / 8), 0, (total - c->conf.n) * (2 * c->sh.rpr / 8),
                               c->stream));
    SOT_TRY(qc_encode_slots(c, points, npoints, (blk_t *)send_dev));
    c->sh.enc_slots_ready = true;
    return SOT_OK;
}

extern "C" int sot_shard_unpack_encoded(sot_ctx *c, const void *recv_dev) {
    SHARD_CTX(c);
    if (!recv_dev) return SOT_EINVAL;
    if (!c->sh.enc_slots_ready) {
        sot_set_error("sot_shard_unpack_encoded before sot_shard_qc_encode");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    const unsigned grid = sgrid(c->sh.n_local);
    const blk_t *r = (const blk_t *)recv_dev;
    BY_WORLD(c, SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_enc<1>, dim3(grid), dim3(256), r, c->sh.chunk, c->sh.n_local, c->role[0].enc, c->role[1].enc),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_enc<2>, dim3(grid), dim3(256), r, c->sh.chunk, c->sh.n_local, c->role[0].enc, c->role[1].enc),
             SOT_LAUNCH(c, KID_SHARD_UNPACK, k_unpack_enc<3>, dim3(grid), dim3(256), r, c->sh.chunk, c->sh.n_local, c->role[0].enc, c->role[1].enc));
    SOT_HIP(hipGetLastError());
    SOT_SYNC(c);
    c->role[0].encoded = c->role[1].encoded = true;
    c->recv_unpacked = false;
    return SOT_OK;
}
