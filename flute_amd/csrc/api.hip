// C ABI of libflute_sot (include/flute_sot.h): configuration, context lifetime, stage entry points.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sot_internal.h"

static thread_local char g_err[512] = "";

void sot_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *sot_last_error(void) { return g_err; }

extern "C" int sot_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; i++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

// ---------------------------------------------------------------- per-kernel event timing
const char *const SOT_KID_NAMES[KID_COUNT] = {
    "k_roots",       "k_ggm_level",     "k_ggm_leaf", "k_sender_treegrp", "k_recv_init",  "k_recv_fix",
    "k_recv_last",   "k_transpose_tiles", "k_sender_hash", "k_receiver_hash", "k_mt_bits",  "k_clear_lsb",
    "k_qc_load",     "k_conv_fwd",      "k_conv_fwd_tile", "k_fft_top",     "k_fft_mid",    "k_fft_low_mul",
    "k_ifft_mid",    "k_ifft_top",      "k_conv_inv_tile", "k_conv_inv",    "k_qc_store",   "k_qc_final",
    "k_and_msg",     "k_and_eval",      "k_shard_pack", "k_shard_unpack"};

static hipEvent_t prof_event(sot_ctx *c) {
    if (!c->prof_free.empty()) {
        hipEvent_t e = c->prof_free.back();
        c->prof_free.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
void prof_begin(sot_ctx *c, int id) {
    prof_rec r;
    r.id = id;
    r.a = prof_event(c);
    r.b = prof_event(c);
    (void)hipEventRecord(r.a, c->stream);
    c->prof_pending.push_back(r);
}
void prof_end(sot_ctx *c, int id) {
    (void)id;
    (void)hipEventRecord(c->prof_pending.back().b, c->stream);
    if (c->prof_pending.size() > 4096) prof_resolve(c);
}
void prof_resolve(sot_ctx *c) {
    if (c->prof_pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto &r : c->prof_pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            c->prof_ms[r.id] += ms;
            c->prof_n[r.id] += 1;
        }
        c->prof_free.push_back(r.a);
        c->prof_free.push_back(r.b);
    }
    c->prof_pending.clear();
}

extern "C" int sot_prof_enable(sot_ctx *c, int on) {
    if (!c) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_resolve(c);
    c->prof = on != 0;
    return SOT_OK;
}
extern "C" int sot_prof_reset(sot_ctx *c) {
    if (!c) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_resolve(c);
    for (int i = 0; i < KID_COUNT; i++) c->prof_ms[i] = 0, c->prof_n[i] = 0;
    return SOT_OK;
}
extern "C" int sot_prof_kernels(void) { return KID_COUNT; }
extern "C" const char *sot_prof_name(int id) { return id >= 0 && id < KID_COUNT ? SOT_KID_NAMES[id] : ""; }
extern "C" int sot_prof_read(sot_ctx *c, int id, double *total_ms, uint64_t *launches) {
    if (!c || id < 0 || id >= KID_COUNT) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    prof_resolve(c);
    if (total_ms) *total_ms = c->prof_ms[id];
    if (launches) *launches = c->prof_n[id];
    return SOT_OK;
}

// ---------------------------------------------------------------- configuration
static int is_prime(uint64_t x) {
    if (x < 2) return 0;
    if (x % 2 == 0) return x == 2;
    for (uint64_t d = 3; d * d <= x; d += 2)
        if (x % d == 0) return 0;
    return 1;
}
// silent_ot/mod.rs:690-695 (same f64 operations)
static uint64_t sec_level(uint64_t scale, uint64_t p, uint64_t points) {
    double x1 = log2((double)(scale * p) / (double)p);
    double x2 = log2((double)(scale * p)) / 2.0;
    return (uint64_t)((double)points * x1 + x2);
}

extern "C" int sot_configure(uint64_t n, uint64_t scaler, uint64_t sec_param, sot_conf *c) {
    if (!c || n == 0 || scaler < 2) {
        sot_set_error("sot_configure: n must be > 0 and scaler >= 2");
        return SOT_EINVAL;
    }
    memset(c, 0, sizeof(*c));
    c->n = n;
    c->scaler = scaler;
    uint64_t m = n > 128 * 128 ? n : 128 * 128;
    uint64_t p = m + 1;  // num_prime next_prime: strictly greater
    while (!is_prime(p)) p++;
    c->P = p;
    uint64_t ret = 1;  // silent_ot/mod.rs:671-681
    while (sec_level(scaler, p, ret) < sec_param) {
        ret++;
        if (ret > 1000) {
            sot_set_error("failed to find silentOt parameters");
            return SOT_EINVAL;
        }
    }
    c->parts = (ret + 7) / 8 * 8;
    uint64_t ss = (p * scaler + c->parts - 1) / c->parts;
    c->size_per = (ss + 7) / 8 * 8;
    c->N2 = c->size_per * c->parts;
    c->N = c->N2 / scaler;
    uint64_t d = 0;
    while ((1ull << d) < c->size_per) d++;
    c->depth = d;
    c->n_blocks = c->N / 128;
    c->n64 = 2 * c->n_blocks;
    c->cols = (c->parts * c->size_per + 127) / 128;
    c->base_ot_count = c->depth * c->parts;
    return SOT_OK;
}

extern "C" size_t sot_treegrp_sums_blocks(const sot_conf *c) { return (c->parts / 8) * 2 * (c->depth - 1) * 8; }
extern "C" size_t sot_treegrp_last_blocks(const sot_conf *c) { return (c->parts / 8) * 8 * 4; }

extern "C" int sot_receiver_points(const sot_conf *c, const uint8_t *choices, uint64_t *points_out) {
    if (!c || !choices || !points_out) return SOT_EINVAL;
    uint64_t num_sets = c->parts / 8;
    for (uint64_t t = 0; t < c->parts; t++) {
        uint64_t p = 0;  // pprf.rs:701-706
        for (uint64_t i = 0; i < c->depth; i++) p |= (uint64_t)(1 ^ choices[t * c->depth + i]) << (c->depth - i - 1);
        // pprf.rs:727-741
        points_out[t] = (t / 8) * 128 + (t % 8 + 8 * (p % 16)) + (p / 16) * num_sets * 128;
    }
    return SOT_OK;
}

#define CHECK_CTX_EARLY(c)                \
    if (!(c)) {                           \
        sot_set_error("null context");    \
        return SOT_EINVAL;                \
    }                                     \
    std::lock_guard<std::recursive_mutex> _lk((c)->mu)

// ---------------------------------------------------------------- context
int dev_alloc(sot_ctx *c, void **p, uint64_t bytes) {
    *p = nullptr;
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) {
        sot_set_error("hipMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
        return SOT_ENOMEM;
    }
    c->dev_bytes += bytes;
    return SOT_OK;
}

int stage_h2d(sot_ctx *c, void *dst_dev, const void *src_host, uint64_t bytes) {
    if (!c->async || bytes > c->h_stage_bytes) {  // a pageable source: the runtime stages it before the call returns
        SOT_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, c->stream));
        if (c->async) SOT_HIP(hipStreamSynchronize(c->stream));
        return SOT_OK;
    }
    const uint64_t need = (bytes + 63) & ~63ull;
    if (c->h_stage_off + need > c->h_stage_bytes) {
        SOT_HIP(hipEventSynchronize(c->stage_done));  // every copy out of the ring so far (recorded behind the last one)
        c->h_stage_off = 0;
    }
    char *h = (char *)c->h_stage + c->h_stage_off;
    memcpy(h, src_host, bytes);
    c->h_stage_off += need;
    SOT_HIP(hipMemcpyAsync(dst_dev, h, bytes, hipMemcpyHostToDevice, c->stream));
    SOT_HIP(hipEventRecord(c->stage_done, c->stream));
    return SOT_OK;
}

extern "C" void sot_ctx_destroy(sot_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    qc_free(c);
    shard_free(c);
    void *ptrs[] = {c->role[0].V, c->role[0].enc, c->role[1].V, c->role[1].enc, c->lvl_a, c->lvl_b, c->sums,
                    c->d_base,    c->d_choices,   c->d_tg_sums, c->d_tg_last,   c->d_rks, c->d_mt,  c->d_tmp,
                    c->d_points,  c->d_chw};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    prof_resolve(c);
    for (hipEvent_t e : c->prof_free) (void)hipEventDestroy(e);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    if (c->stage_done) (void)hipEventDestroy(c->stage_done);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int sot_ctx_create(uint64_t n, uint64_t scaler, int device, sot_ctx **out) {
    if (!out) return SOT_EINVAL;
    *out = nullptr;
    if (scaler < 2 || scaler > 8) {
        sot_set_error("MultType::QuasiCyclic{scaler}: 2 (what bench.rs:85 uses) .. 8 supported, got %llu", (unsigned long long)scaler);
        return SOT_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
        sot_set_error("no HIP device %d (found %d)", device, ndev);
        return SOT_EDEVICE;
    }
    hipDeviceProp_t prop;
    SOT_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        sot_set_error("device %d is %s; libflute_sot is built for gfx950 only", device, prop.gcnArchName);
        return SOT_EDEVICE;
    }
    sot_ctx *c = new sot_ctx();
    c->device = device;
    {
        static const struct {
            const char *name;
            uint32_t bit;
        } sw[] = {{"SOT_NO_TOP", SOT_OPT_NO_TOP},         {"SOT_NO_ASPLIT", SOT_OPT_NO_ASPLIT},   {"SOT_NO_WU", SOT_OPT_NO_WU},
                  {"SOT_NO_W32PAIR", SOT_OPT_NO_W32PAIR}, {"SOT_NO_GGM_TOP", SOT_OPT_NO_GGM_TOP},
                  {"SOT_NO_LUT_COOP", SOT_OPT_NO_LUT_COOP}};
        for (auto &e : sw)
            if (getenv(e.name)) c->opts |= e.bit;
    }
    c->num_cus = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    int rc = sot_configure(n, scaler, 128, &c->conf);
    if (rc != SOT_OK) {
        delete c;
        return rc;
    }
    const sot_conf &cf = c->conf;
    c->nsets = cf.parts / 8;
    c->ggm_set0 = 0, c->ggm_nsets = (uint32_t)c->nsets;
    c->n_out = cf.n;
    if (cf.depth < 4 || cf.parts % 8) {
        sot_set_error("unsupported geometry");
        delete c;
        return SOT_EINVAL;
    }
    rc = hipSetDevice(device) == hipSuccess ? SOT_OK : SOT_EDEVICE;
    if (rc == SOT_OK && hipStreamCreate(&c->own_stream) != hipSuccess) rc = SOT_EDEVICE;
    c->stream = c->own_stream;
    if (rc == SOT_OK && hipEventCreateWithFlags(&c->stage_done, hipEventDisableTiming) != hipSuccess) rc = SOT_EDEVICE;
    uint64_t nw = (cf.n + 63) / 64;
    uint64_t vblocks = cf.cols * 128;
    uint64_t lvl_blocks = cf.parts << (cf.depth - 1);
#define A(ptr, bytes) \
    if (rc == SOT_OK) rc = dev_alloc(c, (void **)&(ptr), (bytes))
    A(c->role[0].V, vblocks * 16);
    A(c->role[1].V, vblocks * 16);
    A(c->role[0].enc, cf.n * 16);
    A(c->role[1].enc, cf.n * 16);
    A(c->lvl_a, lvl_blocks * 16);
    A(c->lvl_b, lvl_blocks * 16);
    A(c->sums, cf.depth * c->nsets * 16 * 16);
    A(c->d_base, cf.base_ot_count * 32);
    A(c->d_choices, cf.base_ot_count);
    A(c->d_tg_sums, sot_treegrp_sums_blocks(&cf) * 16);
    A(c->d_tg_last, sot_treegrp_last_blocks(&cf) * 16);
    A(c->d_mt, 3 * nw * 8);
    A(c->d_chw, nw * 8);
    A(c->d_points, cf.parts * 8);
    // root PRG round keys for up to one worker per tree group (more workers than groups own no group, pprf.rs:128)
    A(c->d_rks, c->nsets * sizeof(aes_rk));
    if (rc == SOT_OK) c->rks_cap = c->nsets;
    // staging for sot_download_rT / sot_qc_dual_encode_host (128 x cols Blocks) and the hashed messages (n x [Block; 2])
    c->tmp_bytes = vblocks * 16 > cf.n * 32 ? vblocks * 16 : cf.n * 32;
    A(c->d_tmp, c->tmp_bytes);
#undef A
    // base OTs + choices + the peer's TreeGrps: receiver_expand's host inputs, staged here when the context is stream-ordered
    c->h_stage_bytes = 4 * (cf.base_ot_count * 17 + (sot_treegrp_sums_blocks(&cf) + sot_treegrp_last_blocks(&cf)) * 16 + cf.parts * 8 + 512);
    if (rc == SOT_OK && hipHostMalloc(&c->h_stage, c->h_stage_bytes, hipHostMallocDefault) != hipSuccess) {
        sot_set_error("hipHostMalloc(%llu bytes) failed", (unsigned long long)c->h_stage_bytes);
        rc = SOT_ENOMEM;
    }
    if (rc == SOT_OK && hipEventRecord(c->stage_done, c->stream) != hipSuccess) rc = SOT_EDEVICE;
    if (rc == SOT_OK) rc = qc_init(c);
    if (rc != SOT_OK) {
        sot_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return SOT_OK;
}

extern "C" int sot_ctx_conf(const sot_ctx *c, sot_conf *out) {
    if (!c || !out) return SOT_EINVAL;
    *out = c->conf;
    return SOT_OK;
}
extern "C" uint64_t sot_ctx_device_bytes(const sot_ctx *c) { return c ? c->dev_bytes : 0; }
extern "C" int sot_ctx_sync(sot_ctx *c) {
    if (!c) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    SOT_HIP(hipSetDevice(c->device));
    SOT_HIP(hipStreamSynchronize(c->stream));
    return SOT_OK;
}
extern "C" void *sot_ctx_stream(sot_ctx *c) { return c ? (void *)c->stream : nullptr; }
extern "C" int sot_ctx_set_stream(sot_ctx *c, void *stream) {
    CHECK_CTX_EARLY(c);
    SOT_HIP(hipSetDevice(c->device));
    prof_resolve(c);
    SOT_HIP(hipStreamSynchronize(c->stream));  // nothing of this context is left on the old stream
    c->stream = stream ? (hipStream_t)stream : c->own_stream;
    return SOT_OK;
}
extern "C" int sot_ctx_set_async(sot_ctx *c, int on) {
    CHECK_CTX_EARLY(c);
    SOT_HIP(hipSetDevice(c->device));
    if (!on) SOT_HIP(hipStreamSynchronize(c->stream));
    c->async = on != 0;
    return SOT_OK;
}

// every entry point that takes a context holds its lock for the whole call (see sot_ctx::mu)
#define CHECK_CTX(c)                      \
    if (!(c)) {                           \
        sot_set_error("null context");    \
        return SOT_EINVAL;                \
    }                                     \
    std::lock_guard<std::recursive_mutex> _lk((c)->mu)
// stage entry points that work on whole leaf matrices: not on a sharded context, whose leaves are spread over ranks
#define CHECK_UNSHARDED(c)                                                                       \
    if ((c)->sh.world > 1) {                                                                     \
        sot_set_error("sharded context: this stage runs through the sot_shard_* entry points"); \
        return SOT_ESTATE;                                                                       \
    }
#define CHECK_ROLE(r)                                  \
    if ((r) != SOT_ROLE_SENDER && (r) != SOT_ROLE_RECEIVER) { \
        sot_set_error("bad role %d", (r));             \
        return SOT_EINVAL;                             \
    }

extern "C" int sot_sender_expand(sot_ctx *c, const void *base_ots, const void *delta, const void *seed,
                                 uint64_t num_threads, void *tg_sums_out, void *tg_last_out) {
    CHECK_CTX(c);
    if (!base_ots || !delta || !seed) {
        sot_set_error("sot_sender_expand: null input");
        return SOT_EINVAL;
    }
    return ggm_sender_expand(c, base_ots, delta, seed, num_threads, tg_sums_out, tg_last_out);
}

extern "C" int sot_receiver_expand(sot_ctx *c, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                                   const void *tg_last) {
    CHECK_CTX(c);
    if (!base_ots || !choices || !tg_sums || !tg_last) {
        sot_set_error("sot_receiver_expand: null input");
        return SOT_EINVAL;
    }
    return ggm_receiver_expand(c, base_ots, choices, tg_sums, tg_last);
}

extern "C" int sot_download_rT(sot_ctx *c, int role, void *rT_out) {
    CHECK_CTX(c);
    CHECK_ROLE(role);
    CHECK_UNSHARDED(c);
    if (!rT_out) return SOT_EINVAL;
    return ggm_download_rT(c, role, rT_out);
}

extern "C" int sot_qc_encode(sot_ctx *c, int role, const uint64_t *points, uint64_t npoints) {
    CHECK_CTX(c);
    CHECK_ROLE(role);
    CHECK_UNSHARDED(c);
    if (!c->role[role].expanded) {
        sot_set_error("qc_encode before expand");
        return SOT_ESTATE;
    }
    if (role == SOT_ROLE_RECEIVER && (!points || npoints != c->conf.parts)) {
        sot_set_error("receiver encode needs the %llu noisy points", (unsigned long long)c->conf.parts);
        return SOT_EINVAL;
    }
    return qc_encode(c, role, points, npoints);
}

extern "C" int sot_qc_encode2(sot_ctx *c, const uint64_t *points, uint64_t npoints, int packing, uint8_t *C_out) {
    CHECK_CTX(c);
    CHECK_UNSHARDED(c);
    if (!c->role[SOT_ROLE_RECEIVER].expanded) {
        sot_set_error("qc_encode2 before receiver expand");
        return SOT_ESTATE;
    }
    if (!points || npoints != c->conf.parts) {
        sot_set_error("receiver encode needs the %llu noisy points", (unsigned long long)c->conf.parts);
        return SOT_EINVAL;
    }
    if (!packing && !C_out) {
        sot_set_error("qc_encode2: ChoiceBitPacking::False returns the choice vector, C_out must not be NULL");
        return SOT_EINVAL;
    }
    return qc_encode(c, SOT_ROLE_RECEIVER, points, npoints, packing ? 1 : 0, C_out);
}

extern "C" int sot_download_encoded(sot_ctx *c, int role, void *blocks_out) {
    CHECK_CTX(c);
    CHECK_ROLE(role);
    if (!blocks_out) {
        sot_set_error("download_encoded: null output");
        return SOT_EINVAL;
    }
    if (!c->role[role].encoded) {
        sot_set_error("download_encoded before qc_encode");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    SOT_HIP(hipMemcpyAsync(blocks_out, c->role[role].enc, c->n_out * 16, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    return SOT_OK;
}

extern "C" int sot_qc_dual_encode_host(sot_ctx *c, int role, const void *rT, const uint64_t *points, uint64_t npoints,
                                       void *blocks_out) {
    CHECK_CTX(c);
    CHECK_ROLE(role);
    CHECK_UNSHARDED(c);
    if (!rT) return SOT_EINVAL;
    SOT_TRY(ggm_upload_rT(c, role, rT));
    SOT_TRY(sot_qc_encode(c, role, points, npoints));
    if (blocks_out) SOT_TRY(sot_download_encoded(c, role, blocks_out));
    return SOT_OK;
}

extern "C" int sot_qc_transform_row(sot_ctx *c, const uint32_t *row_in, int stage, uint32_t *row_out) {
    CHECK_CTX(c);
    CHECK_UNSHARDED(c);
    if (!row_in || !row_out || stage < 0 || stage > 4) return SOT_EINVAL;
    return qc_debug_row(c, row_in, stage, row_out);
}
extern "C" uint32_t sot_qc_log2_capacity(const sot_ctx *c) { return c ? c->g.m : 0; }

extern "C" int sot_sender_hash(sot_ctx *c, const void *delta, void *msgs_out) {
    CHECK_CTX(c);
    if (!delta) return SOT_EINVAL;
    return hash_sender(c, delta, msgs_out);
}
extern "C" int sot_receiver_hash(sot_ctx *c, void *msgs_out, uint64_t *choice_words_out) {
    CHECK_CTX(c);
    return hash_receiver(c, msgs_out, choice_words_out);
}
extern "C" int sot_mt_bits(sot_ctx *c, const void *delta, uint64_t *a, uint64_t *b, uint64_t *cc) {
    CHECK_CTX(c);
    if (!delta) return SOT_EINVAL;
    return hash_mt_bits(c, delta, a, b, cc);
}

extern "C" int sot_mt_role_bits_to_device(sot_ctx *c, int role, const void *delta, void *dst_device, uint64_t dst_bytes) {
    CHECK_CTX(c);
    if ((role != SOT_ROLE_SENDER && role != SOT_ROLE_RECEIVER) || (role == SOT_ROLE_SENDER && !delta)) return SOT_EINVAL;
    const uint64_t bytes = 2 * ((c->n_out + 63) / 64) * 8;
    if (!dst_device || dst_bytes < bytes) {
        sot_set_error("mt_role_bits_to_device: destination of %llu bytes, need %llu", (unsigned long long)dst_bytes,
                      (unsigned long long)bytes);
        return SOT_EINVAL;
    }
    return hash_mt_role_bits(c, role, delta, (uint64_t *)dst_device);
}

extern "C" int sot_mt_bits_to_device(sot_ctx *c, void *dst_device, uint64_t dst_bytes) {
    CHECK_CTX(c);
    const uint64_t bytes = 3 * ((c->n_out + 63) / 64) * 8;
    if (!dst_device || dst_bytes < bytes) {
        sot_set_error("mt_bits_to_device: destination of %llu bytes, need %llu", (unsigned long long)dst_bytes,
                      (unsigned long long)bytes);
        return SOT_EINVAL;
    }
    if (!c->mt_valid) {
        sot_set_error("mt_bits_to_device before sot_mt_bits");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    SOT_HIP(hipMemcpyAsync(dst_device, c->d_mt, bytes, hipMemcpyDeviceToDevice, c->stream));
    SOT_SYNC(c);
    return SOT_OK;
}

extern "C" int sot_blocks_to_device(sot_ctx *c, int what, const void *delta, void *dst_device, uint64_t dst_bytes) {
    CHECK_CTX(c);
    if (what < SOT_OUT_B || what > SOT_OUT_RECEIVER_MSGS) {
        sot_set_error("blocks_to_device: unknown output %d", what);
        return SOT_EINVAL;
    }
    const uint64_t bytes = c->n_out * (what == SOT_OUT_SENDER_MSGS ? 32 : 16);
    if (!dst_device || dst_bytes < bytes) {
        sot_set_error("blocks_to_device: destination of %llu bytes, need %llu", (unsigned long long)dst_bytes,
                      (unsigned long long)bytes);
        return SOT_EINVAL;
    }
    if (what == SOT_OUT_SENDER_MSGS) {
        if (!delta) {
            sot_set_error("blocks_to_device: the sender's hash needs delta");
            return SOT_EINVAL;
        }
        return hash_sender(c, delta, nullptr, (blk_t *)dst_device);
    }
    if (what == SOT_OUT_RECEIVER_MSGS) return hash_receiver(c, nullptr, nullptr, (blk_t *)dst_device);
    const int role = what == SOT_OUT_B ? SOT_ROLE_SENDER : SOT_ROLE_RECEIVER;
    if (!c->role[role].encoded) {
        sot_set_error("blocks_to_device before qc_encode");
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    SOT_HIP(hipMemcpyAsync(dst_device, c->role[role].enc, bytes, hipMemcpyDeviceToDevice, c->stream));
    SOT_SYNC(c);
    return SOT_OK;
}

extern "C" int sot_random_silent_send(sot_ctx *c, const void *base_ots, const void *delta, const void *seed,
                                      uint64_t num_threads, void *tg_sums_out, void *tg_last_out, void *msgs_out) {
    CHECK_CTX(c);
    SOT_TRY(sot_sender_expand(c, base_ots, delta, seed, num_threads, tg_sums_out, tg_last_out));
    SOT_TRY(sot_qc_encode(c, SOT_ROLE_SENDER, nullptr, 0));
    if (msgs_out) SOT_TRY(sot_sender_hash(c, delta, msgs_out));
    return SOT_OK;
}

extern "C" int sot_random_silent_receive(sot_ctx *c, const void *base_ots, const uint8_t *choices, const void *tg_sums,
                                         const void *tg_last, void *msgs_out, uint64_t *choice_words_out) {
    CHECK_CTX(c);
    SOT_TRY(sot_receiver_expand(c, base_ots, choices, tg_sums, tg_last));
    std::vector<uint64_t> pts(c->conf.parts);
    SOT_TRY(sot_receiver_points(&c->conf, choices, pts.data()));
    SOT_TRY(sot_qc_encode(c, SOT_ROLE_RECEIVER, pts.data(), pts.size()));
    if (msgs_out || choice_words_out) SOT_TRY(sot_receiver_hash(c, msgs_out, choice_words_out));
    return SOT_OK;
}

extern "C" int sot_transpose128(int device, const void *in, void *out, uint64_t ntiles) {
    if (!in || !out) return SOT_EINVAL;
    return util_transpose128(device, in, out, ntiles);
}
extern "C" int sot_aes128_encrypt(int device, const void *key16, const void *in, void *out, uint64_t n) {
    if (!key16 || !in || !out) return SOT_EINVAL;
    return util_aes(device, key16, in, out, n, 0);
}
extern "C" int sot_cr_hash(int device, const void *in, void *out, uint64_t n) {
    if (!in || !out) return SOT_EINVAL;
    return util_aes(device, nullptr, in, out, n, 1);
}
