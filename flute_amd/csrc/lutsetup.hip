// FLUTE function-dependent setup on gfx950: the AND layers of the subset-product circuit, bit-packed.
//
// Reference: LutSetupProvider::setup (gmw/src/protocols/aby2_lut.rs:699-783) builds a BooleanGate circuit with one AND
// per subset (size >= 2) of every LUT gate's private input masks (setup_data_circ, :321-370) and runs it through
// Executor<BooleanGmw> (executor.rs:74-193): per layer, split_off_last(count) multiplication triples, compute_msg
// (boolean_gmw.rs:40-67: d = x ^ a, e = y ^ b per gate), one message exchange, evaluate_interactive (:69-95 and
// evaluate.rs:12-21: z = D & b ^ E & a ^ c, party 0 also ^ D & E, with D, E the XOR of both parties' d, e).
//
// Here the host hands over the circuit in flat form (flute_amd/lut_setup.py: operand ids per gate, gates sorted by
// (layer, gate id) = the reference's order inside a layer) and the device keeps everything bit-packed:
//   vals : inputs at bits [0, n_inputs), the outputs of layer L at bits [base[L], base[L] + count[L]), base 64-aligned
//   MTs  : the context's store a | b | c (sot_mt_bits), consumed from the END: layer with `cnt` gates takes bits
//          [mt_rem - cnt, mt_rem) in gate order -- MulTriples::split_off_last (mul_triple/mod.rs:172-185) + iter() (:92-99)
// One thread per gate gathers its two operand bits; a wavefront ballot packs 64 results into one word.
#include <string.h>

#include <vector>

#include "sot_internal.h"

extern "C" void sot_and_circ_destroy(sot_and_circ *q);
struct sot_and_circ {
    sot_ctx *ctx = nullptr;  // must outlive the circuit, except for sot_and_circ_destroy
    int device = 0;
    uint32_t n_inputs = 0, n_gates = 0, n_layers = 0;
    std::vector<uint32_t> count, goff;  // per layer: gates, index of its first gate (execution order)
    std::vector<uint64_t> base;         // per layer: first value bit (64-aligned)
    std::vector<uint64_t> mt0;          // per layer: first MT bit it consumed (set by layer_msg)
    std::vector<uint8_t> msg_done;      // per layer: 1 = message computed, not yet evaluated
    std::vector<uint8_t> evaluated;     // per layer: 1 = outputs written to vals
    int64_t msg_layer = -1;             // the layer whose message is pending (computed, not evaluated); -1: none
    int64_t buf_layer = -1;             // the layer whose message d_own / e_own hold (stays readable for the peer until
                                        // the next layer_msg overwrites it)
    uint64_t val_words = 0, msg_words = 0;
    uint32_t *xpos = nullptr, *ypos = nullptr;  // device: operand bit positions per gate
    uint64_t *vals = nullptr;                   // device
    uint64_t *d_own = nullptr, *e_own = nullptr, *d_peer = nullptr, *e_peer = nullptr;  // device, msg_words each
    // sot_lut_online_eval: device scratch kept across calls (tables of the template; per-item arrays, grown on demand)
    void *le_tab = nullptr;
    uint64_t le_tab_bytes = 0;
    uint8_t *le_items = nullptr;  // masked_in | lam_out | out, le_cap bytes each
    uint64_t le_cap = 0;
    std::vector<uint8_t> le_tab_host;  // what le_tab holds: an unchanged template is not uploaded again
};

__device__ __forceinline__ uint32_t bit_at(const uint64_t *__restrict__ w, uint64_t i) {
    return (uint32_t)(w[i >> 6] >> (i & 63)) & 1u;
}

// BooleanGmw::compute_msg (boolean_gmw.rs:40-67) for the gates goff .. goff+cnt-1 of one layer
__global__ void __launch_bounds__(256)
    k_and_msg(const uint64_t *__restrict__ vals, const uint32_t *__restrict__ xpos, const uint32_t *__restrict__ ypos,
              uint32_t goff, uint32_t cnt, const uint64_t *__restrict__ mta, const uint64_t *__restrict__ mtb,
              uint64_t mt0, uint64_t *__restrict__ dout, uint64_t *__restrict__ eout) {
    const uint32_t cnt_pad = (cnt + 63) & ~63u;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt_pad; j += gridDim.x * blockDim.x) {
        uint32_t d = 0, e = 0;
        if (j < cnt) {
            const uint32_t x = bit_at(vals, xpos[goff + j]), y = bit_at(vals, ypos[goff + j]);
            d = x ^ bit_at(mta, mt0 + j);  // and::compute_shares, evaluate.rs:5-9
            e = y ^ bit_at(mtb, mt0 + j);
        }
        const unsigned long long bd = __ballot(d), be = __ballot(e);
        if ((threadIdx.x & 63) == 0) {
            dout[j >> 6] = bd;
            eout[j >> 6] = be;
        }
    }
}

// BooleanGmw::evaluate_interactive (boolean_gmw.rs:69-95) + and::evaluate (evaluate.rs:12-21)
__global__ void __launch_bounds__(256)
    k_and_eval(const uint64_t *__restrict__ d0, const uint64_t *__restrict__ e0, const uint64_t *__restrict__ d1,
               const uint64_t *__restrict__ e1, const uint64_t *__restrict__ mta, const uint64_t *__restrict__ mtb,
               const uint64_t *__restrict__ mtc, uint64_t mt0, uint32_t cnt, uint32_t party, uint64_t *__restrict__ out) {
    const uint32_t cnt_pad = (cnt + 63) & ~63u;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt_pad; j += gridDim.x * blockDim.x) {
        uint32_t z = 0;
        if (j < cnt) {
            const uint32_t d = bit_at(d0, j) ^ bit_at(d1, j), e = bit_at(e0, j) ^ bit_at(e1, j);
            const uint32_t a = bit_at(mta, mt0 + j), b = bit_at(mtb, mt0 + j), c = bit_at(mtc, mt0 + j);
            z = (d & b) ^ (e & a) ^ c;
            if (party == 0) z ^= d & e;
        }
        const unsigned long long bz = __ballot(z);
        if ((threadIdx.x & 63) == 0) out[j >> 6] = bz;
    }
}

// Both parties' compute_msg + evaluate_interactive of one layer in ONE launch, for in-process two-party runs (both circuits
// on this GPU: tests, benchmarks; sot_and_circ_run_layers_peer). Per gate exactly what k_and_msg and k_and_eval compute
// for party 0 (vals0, triples a0 | b0 | c0 from bit m0) and party 1; the messages are still written out (d0, e0, d1, e1).
struct and_party {
    const uint64_t *vals;
    const uint32_t *xpos, *ypos;
    const uint64_t *mta, *mtb, *mtc;
    uint64_t mt0;
    uint64_t *dout, *eout, *out;
};
__global__ void __launch_bounds__(256) k_and_layer2(and_party p0, and_party p1, uint32_t goff, uint32_t cnt) {
    const uint32_t cnt_pad = (cnt + 63) & ~63u;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < cnt_pad; j += gridDim.x * blockDim.x) {
        uint32_t d0 = 0, e0 = 0, d1 = 0, e1 = 0, z0 = 0, z1 = 0;
        if (j < cnt) {
            const uint32_t a0 = bit_at(p0.mta, p0.mt0 + j), b0 = bit_at(p0.mtb, p0.mt0 + j), c0 = bit_at(p0.mtc, p0.mt0 + j);
            const uint32_t a1 = bit_at(p1.mta, p1.mt0 + j), b1 = bit_at(p1.mtb, p1.mt0 + j), c1 = bit_at(p1.mtc, p1.mt0 + j);
            d0 = bit_at(p0.vals, p0.xpos[goff + j]) ^ a0, e0 = bit_at(p0.vals, p0.ypos[goff + j]) ^ b0;
            d1 = bit_at(p1.vals, p1.xpos[goff + j]) ^ a1, e1 = bit_at(p1.vals, p1.ypos[goff + j]) ^ b1;
            const uint32_t d = d0 ^ d1, e = e0 ^ e1;
            z0 = (d & b0) ^ (e & a0) ^ c0 ^ (d & e);
            z1 = (d & b1) ^ (e & a1) ^ c1;
        }
        const unsigned long long bd0 = __ballot(d0), be0 = __ballot(e0), bd1 = __ballot(d1), be1 = __ballot(e1);
        const unsigned long long bz0 = __ballot(z0), bz1 = __ballot(z1);
        if ((threadIdx.x & 63) == 0) {
            p0.dout[j >> 6] = bd0, p0.eout[j >> 6] = be0, p1.dout[j >> 6] = bd1, p1.eout[j >> 6] = be1;
            p0.out[j >> 6] = bz0, p1.out[j >> 6] = bz1;
        }
    }
}

static inline unsigned lgrid(uint32_t cnt) {
    unsigned g = (cnt + 255) / 256;
    return g < 1 ? 1 : (g > 8192 ? 8192 : g);
}

extern "C" int sot_and_circ_create(sot_ctx *c, uint32_t n_inputs, uint32_t n_gates, uint32_t n_layers,
                                   const uint32_t *layer_counts, const uint32_t *x_src, const uint32_t *y_src,
                                   sot_and_circ **out) {
    if (!c || !out || (n_gates && (!layer_counts || !x_src || !y_src))) return SOT_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    *out = nullptr;
    uint64_t tot = 0;
    for (uint32_t l = 0; l < n_layers; l++) tot += layer_counts[l];
    if (tot != n_gates || (uint64_t)n_inputs + n_gates + 64ull * (n_layers + 1) >= (1ull << 32)) {
        sot_set_error("and_circ: layer counts do not add up to n_gates, or more than 2^32 value bits");
        return SOT_EINVAL;
    }
    sot_and_circ *q = new sot_and_circ();
    q->ctx = c;
    q->device = c->device;
    q->n_inputs = n_inputs, q->n_gates = n_gates, q->n_layers = n_layers;
    q->count.assign(layer_counts, layer_counts + n_layers);
    q->goff.resize(n_layers), q->base.resize(n_layers), q->mt0.assign(n_layers, 0), q->msg_done.assign(n_layers, 0);
    q->evaluated.assign(n_layers, 0);
    uint64_t bit = ((uint64_t)n_inputs + 63) & ~63ull;
    uint32_t g = 0, maxc = 0;
    for (uint32_t l = 0; l < n_layers; l++) {
        q->goff[l] = g, q->base[l] = bit;
        g += q->count[l];
        bit = (bit + q->count[l] + 63) & ~63ull;
        if (q->count[l] > maxc) maxc = q->count[l];
    }
    q->val_words = bit / 64 + 1;
    q->msg_words = ((uint64_t)maxc + 63) / 64 + 1;
    // operand ids -> bit positions; an operand must be an input or a gate of an EARLIER layer
    std::vector<uint32_t> xp(n_gates), yp(n_gates);
    auto pos_of = [&](uint32_t id, uint32_t layer, uint32_t *pos) {
        if (id < n_inputs) {
            *pos = id;
            return true;
        }
        const uint32_t gi = id - n_inputs;
        if (gi >= n_gates) return false;
        uint32_t l = 0;
        while (l + 1 < n_layers && q->goff[l + 1] <= gi) l++;
        if (l >= layer) return false;
        *pos = (uint32_t)(q->base[l] + (gi - q->goff[l]));
        return true;
    };
    for (uint32_t l = 0; l < n_layers; l++)
        for (uint32_t j = 0; j < q->count[l]; j++) {
            const uint32_t gi = q->goff[l] + j;
            if (!pos_of(x_src[gi], l, &xp[gi]) || !pos_of(y_src[gi], l, &yp[gi])) {
                sot_set_error("and_circ: gate %u reads a value that is not an input or an earlier layer's output", gi);
                delete q;
                return SOT_EINVAL;
            }
        }
    int rc = hipSetDevice(c->device) == hipSuccess ? SOT_OK : SOT_EDEVICE;
#define AQ(ptr, bytes) \
    if (rc == SOT_OK && hipMalloc((void **)&(ptr), (bytes)) != hipSuccess) rc = SOT_ENOMEM
    AQ(q->xpos, (uint64_t)(n_gates ? n_gates : 1) * 4);
    AQ(q->ypos, (uint64_t)(n_gates ? n_gates : 1) * 4);
    AQ(q->vals, q->val_words * 8);
    AQ(q->d_own, q->msg_words * 8);
    AQ(q->e_own, q->msg_words * 8);
    AQ(q->d_peer, q->msg_words * 8);
    AQ(q->e_peer, q->msg_words * 8);
#undef AQ
    if (rc == SOT_OK && n_gates) {
        if (hipMemcpy(q->xpos, xp.data(), (uint64_t)n_gates * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(q->ypos, yp.data(), (uint64_t)n_gates * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = SOT_EDEVICE;
    }
    if (rc == SOT_OK && hipMemset(q->vals, 0, q->val_words * 8) != hipSuccess) rc = SOT_EDEVICE;
    if (rc != SOT_OK) {
        sot_set_error("and_circ: device allocation failed");
        sot_and_circ_destroy(q);
        return rc;
    }
    *out = q;
    return SOT_OK;
}

extern "C" void sot_and_circ_destroy(sot_and_circ *q) {
    if (!q) return;
    (void)hipSetDevice(q->device);
    void *ptrs[] = {q->xpos, q->ypos, q->vals, q->d_own, q->e_own, q->d_peer, q->e_peer, q->le_tab, q->le_items};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    delete q;
}

extern "C" int sot_and_circ_set_inputs(sot_and_circ *q, const uint64_t *input_words) {
    if (!q || (!input_words && q->n_inputs)) return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    SOT_HIP(hipSetDevice(c->device));
    SOT_HIP(hipMemsetAsync(q->vals, 0, q->val_words * 8, c->stream));
    if (q->n_inputs) {
        // dead bits of the last word must be zero: they would alias nothing, but keep the image canonical
        std::vector<uint64_t> w(input_words, input_words + ((uint64_t)q->n_inputs + 63) / 64);
        if (q->n_inputs & 63) w.back() &= (1ull << (q->n_inputs & 63)) - 1;
        SOT_HIP(hipMemcpyAsync(q->vals, w.data(), w.size() * 8, hipMemcpyHostToDevice, c->stream));
        SOT_HIP(hipStreamSynchronize(c->stream));
    }
    std::fill(q->msg_done.begin(), q->msg_done.end(), 0);
    std::fill(q->evaluated.begin(), q->evaluated.end(), 0);
    q->msg_layer = -1;
    q->buf_layer = -1;
    return SOT_OK;
}

extern "C" int sot_and_circ_layer_msg(sot_and_circ *q, uint32_t layer, uint8_t *d_out, uint8_t *e_out) {
    if (!q || layer >= q->n_layers) return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    const uint32_t cnt = q->count[layer];
    // the executor's order (executor.rs:120-193): one layer at a time, message then evaluation. There is ONE message
    // buffer, so a second message while one is pending, a layer whose inputs are not evaluated yet, or a repeated
    // layer (its triples would be consumed twice) are refused instead of computing from stale values.
    if (q->msg_layer >= 0) {
        sot_set_error("and_circ: layer_msg(%u) while the message of layer %lld is not evaluated yet", layer,
                      (long long)q->msg_layer);
        return SOT_ESTATE;
    }
    if (q->evaluated[layer] || (layer > 0 && !q->evaluated[layer - 1])) {
        sot_set_error("and_circ: layer_msg(%u) out of order (layers run once each, in ascending order)", layer);
        return SOT_ESTATE;
    }
    if (!c->mt_valid || c->mt_rem < cnt) {
        sot_set_error("and_circ: layer %u needs %u multiplication triples, the store holds %llu", layer, cnt,
                      (unsigned long long)(c->mt_valid ? c->mt_rem : 0));
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nw = (c->n_out + 63) / 64;
    const uint64_t mt0 = c->mt_rem - cnt;  // split_off_last(cnt), executor.rs:139
    SOT_LAUNCH(c, KID_AND_MSG, k_and_msg, dim3(lgrid(cnt)), dim3(256), q->vals, q->xpos, q->ypos, q->goff[layer], cnt,
               c->d_mt, c->d_mt + nw, mt0, q->d_own, q->e_own);
    SOT_HIP(hipGetLastError());
    const uint64_t nb = ((uint64_t)cnt + 7) / 8;
    if (d_out) SOT_HIP(hipMemcpyAsync(d_out, q->d_own, nb, hipMemcpyDeviceToHost, c->stream));
    if (e_out) SOT_HIP(hipMemcpyAsync(e_out, q->e_own, nb, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    // committed only now that the launch and the copies have succeeded
    c->mt_rem = mt0;
    q->mt0[layer] = mt0;
    q->msg_done[layer] = 1;
    q->msg_layer = layer;
    q->buf_layer = layer;
    return SOT_OK;
}

static int and_eval_launch(sot_and_circ *q, uint32_t layer, uint32_t party, const uint64_t *pd, const uint64_t *pe) {
    sot_ctx *c = q->ctx;
    const uint64_t nw = (c->n_out + 63) / 64;
    SOT_LAUNCH(c, KID_AND_EVAL, k_and_eval, dim3(lgrid(q->count[layer])), dim3(256), q->d_own, q->e_own, pd, pe, c->d_mt,
               c->d_mt + nw, c->d_mt + 2 * nw, q->mt0[layer], q->count[layer], party, q->vals + q->base[layer] / 64);
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));
    q->msg_done[layer] = 0;
    q->evaluated[layer] = 1;
    q->msg_layer = -1;
    return SOT_OK;
}

extern "C" int sot_and_circ_layer_eval(sot_and_circ *q, uint32_t layer, uint32_t party_id, const uint8_t *d_peer,
                                       const uint8_t *e_peer) {
    if (!q || layer >= q->n_layers || party_id > 1 || !d_peer || !e_peer) return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!q->msg_done[layer] || q->msg_layer != (int64_t)layer) {
        sot_set_error("and_circ: layer_eval(%u) without this layer's pending layer_msg", layer);
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nb = ((uint64_t)q->count[layer] + 7) / 8;
    SOT_HIP(hipMemsetAsync(q->d_peer, 0, q->msg_words * 8, c->stream));
    SOT_HIP(hipMemsetAsync(q->e_peer, 0, q->msg_words * 8, c->stream));
    SOT_HIP(hipMemcpyAsync(q->d_peer, d_peer, nb, hipMemcpyHostToDevice, c->stream));
    SOT_HIP(hipMemcpyAsync(q->e_peer, e_peer, nb, hipMemcpyHostToDevice, c->stream));
    return and_eval_launch(q, layer, party_id, q->d_peer, q->e_peer);
}

extern "C" int sot_and_circ_layer_eval_peer(sot_and_circ *q, uint32_t layer, uint32_t party_id, sot_and_circ *peer) {
    if (!q || !peer || layer >= q->n_layers || party_id > 1 || peer->n_layers != q->n_layers ||
        peer->count[layer] != q->count[layer] || peer->ctx->device != q->ctx->device)
        return SOT_EINVAL;
    // both contexts: the peer's message buffers are read on this context's stream (address order: no deadlock when the
    // two parties evaluate each other's layers from two threads)
    sot_ctx *ca = q->ctx < peer->ctx ? q->ctx : peer->ctx, *cb = q->ctx < peer->ctx ? peer->ctx : q->ctx;
    std::lock_guard<std::recursive_mutex> lk1(ca->mu);
    std::lock_guard<std::recursive_mutex> lk2(cb->mu);
    // the peer's message stays readable until ITS next layer_msg (buf_layer): it may already have evaluated this layer
    if (!q->msg_done[layer] || q->msg_layer != (int64_t)layer || peer->buf_layer != (int64_t)layer) {
        sot_set_error("and_circ: layer_eval_peer(%u) before both parties' layer_msg of this layer", layer);
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(q->ctx->device));
    // the peer's message is copied out of its buffers (it may start its next layer at once); its own stream has been
    // synchronised by its layer_msg, so the bytes are complete
    const uint64_t nbw = (((uint64_t)q->count[layer] + 63) / 64) * 8;
    SOT_HIP(hipMemcpyAsync(q->d_peer, peer->d_own, nbw, hipMemcpyDeviceToDevice, q->ctx->stream));
    SOT_HIP(hipMemcpyAsync(q->e_peer, peer->e_own, nbw, hipMemcpyDeviceToDevice, q->ctx->stream));
    return and_eval_launch(q, layer, party_id, q->d_peer, q->e_peer);
}

// evaluate `layer` with the peer's message, then compute the message of layer + 1 at once: ONE host round trip per
// network round instead of two (the next message depends on nothing but this layer's outputs, executor.rs:139-163)
extern "C" int sot_and_circ_layer_eval_msg(sot_and_circ *q, uint32_t layer, uint32_t party_id, const uint8_t *d_peer,
                                           const uint8_t *e_peer, uint8_t *d_next_out, uint8_t *e_next_out) {
    if (!q || layer >= q->n_layers || party_id > 1 || !d_peer || !e_peer) return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    if (!q->msg_done[layer] || q->msg_layer != (int64_t)layer) {
        sot_set_error("and_circ: layer_eval_msg(%u) without this layer's pending layer_msg", layer);
        return SOT_ESTATE;
    }
    const bool more = layer + 1 < q->n_layers;
    const uint32_t cnt1 = more ? q->count[layer + 1] : 0;
    if (more && (!c->mt_valid || c->mt_rem < cnt1)) {
        sot_set_error("and_circ: layer %u needs %u multiplication triples, the store holds %llu", layer + 1, cnt1,
                      (unsigned long long)(c->mt_valid ? c->mt_rem : 0));
        return SOT_ESTATE;
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t nw = (c->n_out + 63) / 64;
    const uint64_t nb = ((uint64_t)q->count[layer] + 7) / 8;
    SOT_HIP(hipMemsetAsync(q->d_peer, 0, q->msg_words * 8, c->stream));
    SOT_HIP(hipMemsetAsync(q->e_peer, 0, q->msg_words * 8, c->stream));
    SOT_HIP(hipMemcpyAsync(q->d_peer, d_peer, nb, hipMemcpyHostToDevice, c->stream));
    SOT_HIP(hipMemcpyAsync(q->e_peer, e_peer, nb, hipMemcpyHostToDevice, c->stream));
    SOT_LAUNCH(c, KID_AND_EVAL, k_and_eval, dim3(lgrid(q->count[layer])), dim3(256), q->d_own, q->e_own, q->d_peer, q->e_peer,
               c->d_mt, c->d_mt + nw, c->d_mt + 2 * nw, q->mt0[layer], q->count[layer], party_id, q->vals + q->base[layer] / 64);
    uint64_t mt0 = 0;
    if (more) {  // stream order: the evaluation above has written this layer's outputs before the message kernel reads them
        mt0 = c->mt_rem - cnt1;
        SOT_LAUNCH(c, KID_AND_MSG, k_and_msg, dim3(lgrid(cnt1)), dim3(256), q->vals, q->xpos, q->ypos, q->goff[layer + 1], cnt1,
                   c->d_mt, c->d_mt + nw, mt0, q->d_own, q->e_own);
        const uint64_t nb1 = ((uint64_t)cnt1 + 7) / 8;
        if (d_next_out) SOT_HIP(hipMemcpyAsync(d_next_out, q->d_own, nb1, hipMemcpyDeviceToHost, c->stream));
        if (e_next_out) SOT_HIP(hipMemcpyAsync(e_next_out, q->e_own, nb1, hipMemcpyDeviceToHost, c->stream));
    }
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c->stream));
    q->msg_done[layer] = 0, q->evaluated[layer] = 1, q->msg_layer = -1;
    if (more) {
        c->mt_rem = mt0;
        q->mt0[layer + 1] = mt0, q->msg_done[layer + 1] = 1, q->msg_layer = layer + 1, q->buf_layer = layer + 1;
    }
    return SOT_OK;
}

// Both parties' layers first .. last in one call for in-process two-party runs (circ0 = party 0, circ1 = party 1, both
// on the same GPU; the reference's analogue is two executors over mpc_channel::in_memory): one fused launch per layer,
// queued on party 0's stream, ONE host synchronisation at the end. Each party's triples come from the end of its own
// context's store, exactly as layer_msg takes them.
extern "C" int sot_and_circ_run_layers_peer(sot_and_circ *q0, sot_and_circ *q1, uint32_t first, uint32_t last) {
    if (!q0 || !q1 || q0 == q1 || first > last || last >= q0->n_layers || q0->n_layers != q1->n_layers ||
        q0->n_gates != q1->n_gates || q0->n_inputs != q1->n_inputs || q0->ctx->device != q1->ctx->device ||
        q0->ctx == q1->ctx) {
        sot_set_error("and_circ: run_layers_peer needs two circuits of the same shape on two contexts of one GPU and first <= last < n_layers");
        return SOT_EINVAL;
    }
    for (uint32_t l = first; l <= last; l++)
        if (q0->count[l] != q1->count[l]) return SOT_EINVAL;
    sot_ctx *c0 = q0->ctx, *c1 = q1->ctx;
    sot_ctx *ca = c0 < c1 ? c0 : c1, *cb = c0 < c1 ? c1 : c0;
    std::lock_guard<std::recursive_mutex> lk1(ca->mu);
    std::lock_guard<std::recursive_mutex> lk2(cb->mu);
    uint64_t need = 0;
    for (uint32_t l = first; l <= last; l++) need += q0->count[l];
    for (sot_and_circ *q : {q0, q1}) {
        if (q->msg_layer >= 0 || q->evaluated[first] || (first > 0 && !q->evaluated[first - 1])) {
            sot_set_error("and_circ: run_layers_peer(%u..%u) out of order (a message is pending, or the layers before are "
                          "not evaluated, or layer %u already is)", first, last, first);
            return SOT_ESTATE;
        }
        if (!q->ctx->mt_valid || q->ctx->mt_rem < need) {
            sot_set_error("and_circ: layers %u..%u need %llu multiplication triples, the store holds %llu", first, last,
                          (unsigned long long)need, (unsigned long long)(q->ctx->mt_valid ? q->ctx->mt_rem : 0));
            return SOT_ESTATE;
        }
    }
    SOT_HIP(hipSetDevice(c0->device));
    if (c1 != c0) SOT_HIP(hipStreamSynchronize(c1->stream));  // party 1's store and values are complete (normally idle already)
    const uint64_t nw0 = (c0->n_out + 63) / 64, nw1 = (c1->n_out + 63) / 64;
    uint64_t rem0 = c0->mt_rem, rem1 = c1->mt_rem;
    for (uint32_t l = first; l <= last; l++) {
        const uint32_t cnt = q0->count[l];
        rem0 -= cnt, rem1 -= cnt;
        and_party p0 = {q0->vals, q0->xpos, q0->ypos, c0->d_mt, c0->d_mt + nw0, c0->d_mt + 2 * nw0, rem0,
                        q0->d_own, q0->e_own, q0->vals + q0->base[l] / 64};
        and_party p1 = {q1->vals, q1->xpos, q1->ypos, c1->d_mt, c1->d_mt + nw1, c1->d_mt + 2 * nw1, rem1,
                        q1->d_own, q1->e_own, q1->vals + q1->base[l] / 64};
        SOT_LAUNCH(c0, KID_AND_EVAL, k_and_layer2, dim3(lgrid(cnt)), dim3(256), p0, p1, q0->goff[l], cnt);
        q0->mt0[l] = rem0, q1->mt0[l] = rem1;
    }
    SOT_HIP(hipGetLastError());
    SOT_HIP(hipStreamSynchronize(c0->stream));
    c0->mt_rem = rem0, c1->mt_rem = rem1;
    for (sot_and_circ *q : {q0, q1}) {
        for (uint32_t l = first; l <= last; l++) q->msg_done[l] = 0, q->evaluated[l] = 1;
        q->msg_layer = -1, q->buf_layer = last;
    }
    return SOT_OK;
}

extern "C" int sot_and_circ_values(sot_and_circ *q, uint8_t *bits_out) {
    if (!q || !bits_out) return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    SOT_HIP(hipSetDevice(c->device));
    std::vector<uint64_t> w(q->val_words);
    SOT_HIP(hipMemcpyAsync(w.data(), q->vals, q->val_words * 8, hipMemcpyDeviceToHost, c->stream));
    SOT_HIP(hipStreamSynchronize(c->stream));
    auto bit = [&](uint64_t i) { return (uint8_t)((w[i >> 6] >> (i & 63)) & 1); };
    for (uint32_t i = 0; i < q->n_inputs; i++) bits_out[i] = bit(i);
    for (uint32_t l = 0; l < q->n_layers; l++)
        for (uint32_t j = 0; j < q->count[l]; j++) bits_out[q->n_inputs + q->goff[l] + j] = bit(q->base[l] + j);
    return SOT_OK;
}

extern "C" uint64_t sot_mt_remaining(const sot_ctx *c) { return c && c->mt_valid ? c->mt_rem : 0; }

// ================================================================================================
// FLUTE ONLINE LUT evaluation: LutGate::compute_delta_share (aby2_lut.rs:209-277) for `batch` SIMD
// instances of a template of LUT gates (flute_benchmark batches sub-circuits this way, bench.rs:367-384), on the subset
// products the setup circuit above left in HBM.
//
// For one gate with p inputs, truth table T (row index j: input i is bit p-1-i of j, aby2_lut.rs:959), public masked
// inputs m and shared lambdas, the reference XORs, over the subsets q of the inputs, [XOR over the set rows j of
// AND_{i in q} x_i[j]] & share(prod_{i not in q} lambda_i), x_i[j] = !m_i ^ j_i (expand, :947-964). AND_{i in q} x_i[j] = 1
// iff j agrees with m on q, so with g(j') = T[j' ^ m] the bracket is Z[~q] for Z = the subset-sum (zeta) transform of g
// over GF(2), and the party's share of the output value is parity(Z & L), L[s] = its share of prod_{i in s} lambda_i
// (L[0] = 1 for party 1 only: the reference's whole_set_mask, :265-270). Per thread: one XOR-permutation of the table by
// m, one 2^p-point zeta transform on <= 8 words, 2^p - 1 share bits gathered from the value store.
// The reference's bitvec_and_fold (:1177-1188) folds from all-false, which zeroes every term (its tests are #[ignore]d);
// this computes the protocol as specified -- see oracle/lut_eval_oracle.py for the restatement it is tested against.
//
// lsrc[g][s], 0 < s < 2^p: where L[s] lives for template gate g: 0 << 24 | i = setup input i of the instance
// (|s| = 1), (l + 1) << 24 | r = the gate of rank r among the template's gates of setup layer l.
__global__ void __launch_bounds__(256)
    k_lut_online(const uint64_t *__restrict__ vals, uint32_t inst0, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                 const uint32_t *__restrict__ arity, const uint32_t *__restrict__ tt, const uint32_t *__restrict__ lsrc,
                 const uint64_t *__restrict__ lbase, const uint32_t *__restrict__ tcount,
                 const uint8_t *__restrict__ masked_in, const uint8_t *__restrict__ lam_out, uint32_t party,
                 uint8_t *__restrict__ out) {
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint64_t)batch * n_luts) return;
    const uint32_t inst = inst0 + (uint32_t)(idx / n_luts), g = (uint32_t)(idx % n_luts);  // instance of the setup circuit
    const uint32_t p = arity[g], nrow = 1u << p, nw = nrow > 32 ? nrow >> 5 : 1;
    const uint32_t m = masked_in[idx] & (nrow - 1u);
    uint32_t G[8];
#pragma unroll
    for (uint32_t w = 0; w < 8; w++) G[w] = w < nw ? tt[g * 8 + (w ^ (m >> 5))] : 0u;
    // g(j') = T[j' ^ m] inside the words: swap the halves of every 2^(b+1)-block for the set bits b of m
#pragma unroll
    for (uint32_t b = 0; b < 5; b++) {
        const uint32_t M = b == 0 ? 0x55555555u : b == 1 ? 0x33333333u : b == 2 ? 0x0F0F0F0Fu : b == 3 ? 0x00FF00FFu : 0x0000FFFFu;
        if ((m >> b) & 1u) {
#pragma unroll
            for (uint32_t w = 0; w < 8; w++) G[w] = ((G[w] & M) << (1u << b)) | ((G[w] >> (1u << b)) & M);
        }
    }
    // Z[s] = XOR of g over the subsets of s
#pragma unroll
    for (uint32_t b = 0; b < 5; b++) {
        const uint32_t M = b == 0 ? 0x55555555u : b == 1 ? 0x33333333u : b == 2 ? 0x0F0F0F0Fu : b == 3 ? 0x00FF00FFu : 0x0000FFFFu;
        if (b < p) {
#pragma unroll
            for (uint32_t w = 0; w < 8; w++) G[w] ^= (G[w] & M) << (1u << b);
        }
    }
#pragma unroll
    for (uint32_t wb = 0; wb < 3; wb++) {
        if (5 + wb < p) {
#pragma unroll
            for (uint32_t w = 0; w < 8; w++)
                if ((w >> wb) & 1u) G[w] ^= G[w ^ (1u << wb)];
        }
    }
    uint32_t acc = G[0] & party;  // L[0]
    const uint32_t *src = lsrc + (uint64_t)g * 256;
#pragma unroll
    for (uint32_t w = 0; w < 8; w++) {
        if (w < nw) {
            const uint32_t z = G[w];
            const uint32_t hi = nrow < 32 ? nrow : 32;
#pragma unroll 4
            for (uint32_t b = (w == 0 ? 1 : 0); b < hi; b++) {
                const uint32_t e = src[w * 32 + b], l1 = e >> 24, r = e & 0xFFFFFFu;
                const uint64_t pos = l1 == 0 ? (uint64_t)inst * n_in + r : lbase[l1 - 1] + (uint64_t)inst * tcount[l1 - 1] + r;
                acc ^= (z >> b) & bit_at(vals, pos);
            }
        }
    }
    out[idx] = (uint8_t)((acc ^ lam_out[idx]) & 1u);
}

// The same computation, workgroup-cooperative (round 3): a workgroup takes `ipb` consecutive instances and
//   1. stages each instance's setup shares -- its n_in input bits and its tcount[l] gate bits of every setup layer, each a
//      contiguous bit range of the value store -- into LDS as a compact vector (segments padded to whole words: woff);
//   2. assembles the 2^p-bit vector L (L[s] = share of the product over subset s, L[0] = party) ONCE per (instance, class),
//      a class being the template gates with the same arity and the same lsrc row (all 8 LUT gates of sbox8.lut: one input
//      set, one cached subset circuit) -- the words of L are dealt to the class's gates, so a gate gathers 2^p / count bits
//      from LDS instead of 2^p - 1 bits from HBM;
//   3. per (instance, gate): the table permuted by m, its zeta transform Z, parity(Z & L) ^ lambda_out.
// cpos[g][s]: compact bit position of lsrc[g][s]; gcls[g] = class | rank << 16 | min(count, 8) << 24 (rank 255: no words).
// bits [bit, bit + need) of the store, need <= 32 (the following word is only touched when those bits reach into it: the last
// segment of the last instance ends with the allocation)
__device__ __forceinline__ uint32_t bits32_at(const uint64_t *__restrict__ w, uint64_t bit, uint32_t need) {
    const uint32_t sh = (uint32_t)(bit & 63);
    uint64_t v = w[bit >> 6] >> sh;
    if (sh + need > 64) v |= w[(bit >> 6) + 1] << (64 - sh);
    return (uint32_t)v;
}
__global__ void __launch_bounds__(256)
    k_lut_online2(const uint64_t *__restrict__ vals, uint32_t inst0, uint32_t batch, uint32_t n_luts, uint32_t n_in, uint32_t nlay,
                  const uint32_t *__restrict__ arity, const uint32_t *__restrict__ tt, const uint32_t *__restrict__ cpos,
                  const uint32_t *__restrict__ gcls, uint32_t ncls, const uint64_t *__restrict__ lbase,
                  const uint32_t *__restrict__ tcount, const uint32_t *__restrict__ woff, uint32_t CW, uint32_t ipb,
                  const uint8_t *__restrict__ masked_in, const uint8_t *__restrict__ lam_out, uint32_t party,
                  uint8_t *__restrict__ out) {
    extern __shared__ uint32_t lo_sm[];
    uint32_t *comp = lo_sm, *Lb = lo_sm + ipb * CW;
    const uint32_t i0 = blockIdx.x * ipb;
    const uint32_t ni = batch - i0 < ipb ? batch - i0 : ipb;
    for (uint32_t it = threadIdx.x; it < ni * CW; it += 256) {
        const uint32_t il = it / CW, cw = it - il * CW;
        uint32_t sg = 0;
        while (sg < nlay && woff[sg + 1] <= cw) sg++;
        const uint32_t c = cw - woff[sg], len = (sg == 0 ? n_in : tcount[sg - 1]) - 32 * c;
        const uint64_t inst = (uint64_t)inst0 + i0 + il;
        const uint64_t bit = (sg == 0 ? inst * n_in : lbase[sg - 1] + inst * tcount[sg - 1]) + 32 * c;
        uint32_t v = bits32_at(vals, bit, len < 32 ? len : 32u);
        if (len < 32) v &= (1u << len) - 1u;
        comp[it] = v;
    }
    __syncthreads();
    const uint32_t items = ni * n_luts;
    for (uint32_t it = threadIdx.x; it < items; it += 256) {
        const uint32_t il = it / n_luts, g = it - il * n_luts;
        const uint32_t info = gcls[g], cl = info & 0xFFFFu, rk = (info >> 16) & 0xFFu, cn = info >> 24;
        if (rk == 255u) continue;
        const uint32_t p = arity[g], nrow = 1u << p, nw = nrow > 32 ? nrow >> 5 : 1, hi = nrow < 32 ? nrow : 32;
        const uint32_t *cp = cpos + (uint64_t)g * 256;
        const uint32_t *cv = comp + il * CW;
        for (uint32_t w = rk; w < nw; w += cn) {
            uint32_t word = w == 0 ? (party & 1u) : 0u;
            for (uint32_t b = (w == 0 ? 1 : 0); b < hi; b++) {
                const uint32_t q = cp[w * 32 + b];
                word |= ((cv[q >> 5] >> (q & 31)) & 1u) << b;
            }
            Lb[(il * ncls + cl) * 8 + w] = word;
        }
    }
    __syncthreads();
    for (uint32_t it = threadIdx.x; it < items; it += 256) {
        const uint32_t il = it / n_luts, g = it - il * n_luts;
        const uint64_t idx = (uint64_t)(i0 + il) * n_luts + g;
        const uint32_t p = arity[g], nrow = 1u << p, nw = nrow > 32 ? nrow >> 5 : 1;
        const uint32_t m = masked_in[idx] & (nrow - 1u);
        uint32_t G[8];
#pragma unroll
        for (uint32_t w = 0; w < 8; w++) G[w] = w < nw ? tt[g * 8 + (w ^ (m >> 5))] : 0u;
#pragma unroll
        for (uint32_t b = 0; b < 5; b++) {
            const uint32_t M = b == 0 ? 0x55555555u : b == 1 ? 0x33333333u : b == 2 ? 0x0F0F0F0Fu : b == 3 ? 0x00FF00FFu : 0x0000FFFFu;
            if ((m >> b) & 1u) {
#pragma unroll
                for (uint32_t w = 0; w < 8; w++) G[w] = ((G[w] & M) << (1u << b)) | ((G[w] >> (1u << b)) & M);
            }
        }
#pragma unroll
        for (uint32_t b = 0; b < 5; b++) {
            const uint32_t M = b == 0 ? 0x55555555u : b == 1 ? 0x33333333u : b == 2 ? 0x0F0F0F0Fu : b == 3 ? 0x00FF00FFu : 0x0000FFFFu;
            if (b < p) {
#pragma unroll
                for (uint32_t w = 0; w < 8; w++) G[w] ^= (G[w] & M) << (1u << b);
            }
        }
#pragma unroll
        for (uint32_t wb = 0; wb < 3; wb++) {
            if (5 + wb < p) {
#pragma unroll
                for (uint32_t w = 0; w < 8; w++)
                    if ((w >> wb) & 1u) G[w] ^= G[w ^ (1u << wb)];
            }
        }
        const uint32_t *L = Lb + (il * ncls + (gcls[g] & 0xFFFFu)) * 8;
        const uint32_t live = nrow < 32 ? (1u << nrow) - 1u : 0xFFFFFFFFu;
        uint32_t acc = 0;
#pragma unroll
        for (uint32_t w = 0; w < 8; w++)
            if (w < nw) acc ^= G[w] & L[w] & live;
        out[idx] = (uint8_t)((__popc(acc) ^ lam_out[idx]) & 1u);
    }
}

// grows a device buffer kept in the circuit (no allocation once it is large enough)
static int le_reserve(void **buf, uint64_t *cap, uint64_t bytes) {
    if (*cap >= bytes && *buf) return SOT_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr, *cap = 0;
    if (hipMalloc(buf, bytes ? bytes : 16) != hipSuccess) {
        (void)hipGetLastError();
        sot_set_error("lut_online: hipMalloc(%llu bytes) failed", (unsigned long long)bytes);
        return SOT_ENOMEM;
    }
    *cap = bytes;
    return SOT_OK;
}

static int lut_online(sot_and_circ *q, uint32_t first_instance, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                      const uint32_t *arity, const uint32_t *tt, const uint32_t *lsrc, const uint32_t *tcount,
                      const uint8_t *masked_in, const uint8_t *lam_out, uint32_t party_id, uint8_t *delta_share_out,
                      bool on_device) {
    if (!q || !batch || !n_luts || !arity || !tt || !lsrc || (!tcount && q->n_layers) || !masked_in || !lam_out ||
        party_id > 1 || !delta_share_out)
        return SOT_EINVAL;
    sot_ctx *c = q->ctx;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    // the store must hold the evaluated setup circuit of N instances of the template, N >= first_instance + batch
    const uint64_t ninst = n_in ? q->n_inputs / n_in : 0;
    if (!n_in || ninst * n_in != q->n_inputs || (uint64_t)first_instance + batch > ninst) {
        sot_set_error("lut_online: instances %u .. %u of a setup circuit with %u inputs = %llu instances x %u", first_instance,
                      first_instance + batch, q->n_inputs, (unsigned long long)ninst, n_in);
        return SOT_EINVAL;
    }
    for (uint32_t l = 0; l < q->n_layers; l++) {
        if (ninst * tcount[l] != q->count[l]) {
            sot_set_error("lut_online: setup layer %u has %u gates, not %llu instances x %u", l, q->count[l],
                          (unsigned long long)ninst, tcount[l]);
            return SOT_EINVAL;
        }
        if (!q->evaluated[l]) {
            sot_set_error("lut_online: setup layer %u is not evaluated yet", l);
            return SOT_ESTATE;
        }
    }
    for (uint32_t g = 0; g < n_luts; g++) {
        if (arity[g] < 1 || arity[g] > 8) {
            sot_set_error("lut_online: LUT %u has %u inputs (1 .. 8 supported)", g, arity[g]);
            return SOT_EINVAL;
        }
        for (uint32_t s = 1; s < (1u << arity[g]); s++) {
            const uint32_t e = lsrc[(uint64_t)g * 256 + s], l1 = e >> 24, r = e & 0xFFFFFFu;
            if (l1 == 0 ? r >= n_in : (l1 > q->n_layers || r >= tcount[l1 - 1])) {
                sot_set_error("lut_online: LUT %u subset %u points outside the setup circuit", g, s);
                return SOT_EINVAL;
            }
        }
    }
    SOT_HIP(hipSetDevice(c->device));
    const uint64_t items = (uint64_t)batch * n_luts;
    // tables: arity | tt | lsrc | tcount | layer bases | cpos | gcls | woff, 8-byte aligned pieces in one buffer
    const uint32_t nlay = q->n_layers;
    const uint64_t o_ar = 0, o_tt = o_ar + (((uint64_t)n_luts * 4 + 7) & ~7ull), o_ls = o_tt + (uint64_t)n_luts * 32;
    const uint64_t o_tc = o_ls + (uint64_t)n_luts * 1024, o_lb = o_tc + (((uint64_t)nlay * 4 + 7) & ~7ull);
    const uint64_t o_cp = o_lb + (uint64_t)nlay * 8 + 8, o_gc = o_cp + (uint64_t)n_luts * 1024;
    const uint64_t o_wo = o_gc + (((uint64_t)n_luts * 4 + 7) & ~7ull);
    const uint64_t tab_bytes = o_wo + (((uint64_t)(nlay + 2) * 4 + 7) & ~7ull);
    SOT_TRY(le_reserve(&q->le_tab, &q->le_tab_bytes, tab_bytes));
    // compact layout of one instance's shares: the inputs, then every setup layer's gates, each segment padded to words
    std::vector<uint32_t> woff(nlay + 2, 0);
    woff[1] = (n_in + 31) / 32;
    for (uint32_t l = 0; l < nlay; l++) woff[l + 2] = woff[l + 1] + (tcount[l] + 31) / 32;
    const uint32_t CW = woff[nlay + 1];
    uint32_t ncls = 0;
    {
        // one host image of the tables; uploaded only when it differs from what the device copy holds (a deeper circuit
        // calls once per LUT layer with the same template: no PCIe traffic but the per-item arrays)
        std::vector<uint8_t> img(tab_bytes, 0);
        memcpy(img.data() + o_ar, arity, (uint64_t)n_luts * 4);
        memcpy(img.data() + o_tt, tt, (uint64_t)n_luts * 32);
        memcpy(img.data() + o_ls, lsrc, (uint64_t)n_luts * 1024);
        if (nlay) {
            memcpy(img.data() + o_tc, tcount, (uint64_t)nlay * 4);
            memcpy(img.data() + o_lb, q->base.data(), (uint64_t)nlay * 8);
        }
        uint32_t *cp = (uint32_t *)(img.data() + o_cp), *gc = (uint32_t *)(img.data() + o_gc);
        memcpy(img.data() + o_wo, woff.data(), (uint64_t)(nlay + 2) * 4);
        std::vector<uint32_t> rep, cnt;  // per class: a representative gate, gates so far
        for (uint32_t g = 0; g < n_luts; g++) {
            const uint32_t rows = 1u << arity[g];
            for (uint32_t sx = 1; sx < rows; sx++) {
                const uint32_t e = lsrc[(uint64_t)g * 256 + sx], l1 = e >> 24, r = e & 0xFFFFFFu;
                cp[(uint64_t)g * 256 + sx] = 32 * woff[l1] + r;
            }
            uint32_t cl = 0;
            for (; cl < rep.size(); cl++)
                if (arity[rep[cl]] == arity[g] && !memcmp(lsrc + (uint64_t)rep[cl] * 256 + 1, lsrc + (uint64_t)g * 256 + 1, (rows - 1) * 4)) break;
            if (cl == rep.size()) rep.push_back(g), cnt.push_back(0);
            gc[g] = cl | ((cnt[cl] < 8 ? cnt[cl] : 255u) << 16);
            cnt[cl]++;
        }
        ncls = (uint32_t)rep.size();
        for (uint32_t g = 0; g < n_luts; g++) gc[g] |= (cnt[gc[g] & 0xFFFFu] < 8 ? cnt[gc[g] & 0xFFFFu] : 8u) << 24;
        if (img != q->le_tab_host) {
            SOT_HIP(hipStreamSynchronize(c->stream));  // a kernel of the previous call may still read the old tables
            SOT_HIP(hipMemcpy(q->le_tab, img.data(), tab_bytes, hipMemcpyHostToDevice));
            q->le_tab_host.swap(img);
        }
    }
    char *tb = (char *)q->le_tab;
    const uint32_t *d_ar = (const uint32_t *)(tb + o_ar), *d_tt = (const uint32_t *)(tb + o_tt), *d_ls = (const uint32_t *)(tb + o_ls);
    const uint32_t *d_tc = (const uint32_t *)(tb + o_tc);
    const uint64_t *d_lb = (const uint64_t *)(tb + o_lb);
    const uint8_t *d_m = masked_in, *d_lo = lam_out;
    uint8_t *d_out = delta_share_out;
    if (!on_device) {
        if (q->le_cap < items) {  // three byte arrays of le_cap entries each
            uint64_t got = 0;
            void *p = q->le_items;
            q->le_items = nullptr, q->le_cap = 0;
            SOT_HIP(hipStreamSynchronize(c->stream));
            SOT_TRY(le_reserve(&p, &got, 3 * items));
            q->le_items = (uint8_t *)p, q->le_cap = items;
        }
        SOT_HIP(hipMemcpyAsync(q->le_items, masked_in, items, hipMemcpyHostToDevice, c->stream));
        SOT_HIP(hipMemcpyAsync(q->le_items + q->le_cap, lam_out, items, hipMemcpyHostToDevice, c->stream));
        d_m = q->le_items, d_lo = q->le_items + q->le_cap, d_out = q->le_items + 2 * q->le_cap;
    }
    // workgroup-cooperative kernel when `ipb` instances' compact shares and L vectors fit 48 KiB of LDS (any template of a
    // few hundred gates); otherwise one thread per item gathering from HBM
    uint32_t ipb = n_luts >= 256 ? 1 : 256 / n_luts;
    while (ipb > 1 && (uint64_t)ipb * (CW + 8ull * ncls) * 4 > 48 * 1024) ipb /= 2;
    if ((uint64_t)ipb * (CW + 8ull * ncls) * 4 <= 48 * 1024 && ncls < 65536 && !(c->opts & SOT_OPT_NO_LUT_COOP)) {
        hipLaunchKernelGGL(k_lut_online2, dim3((batch + ipb - 1) / ipb), dim3(256), ipb * (CW + 8 * ncls) * 4, c->stream, q->vals,
                           first_instance, batch, n_luts, n_in, nlay, d_ar, d_tt, (const uint32_t *)(tb + o_cp),
                           (const uint32_t *)(tb + o_gc), ncls, d_lb, d_tc, (const uint32_t *)(tb + o_wo), CW, ipb, d_m, d_lo,
                           party_id, d_out);
    } else {
        const uint64_t grid = (items + 255) / 256;
        hipLaunchKernelGGL(k_lut_online, dim3((unsigned)grid), dim3(256), 0, c->stream, q->vals, first_instance, batch, n_luts, n_in,
                           d_ar, d_tt, d_ls, d_lb, d_tc, d_m, d_lo, party_id, d_out);
    }
    SOT_HIP(hipGetLastError());
    if (!on_device) {
        SOT_HIP(hipMemcpyAsync(delta_share_out, d_out, items, hipMemcpyDeviceToHost, c->stream));
        SOT_HIP(hipStreamSynchronize(c->stream));
    }
    return SOT_OK;
}

extern "C" int sot_lut_online_eval(sot_and_circ *q, uint32_t first_instance, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                                   const uint32_t *arity, const uint32_t *tt, const uint32_t *lsrc, const uint32_t *tcount,
                                   const uint8_t *masked_in, const uint8_t *lam_out, uint32_t party_id,
                                   uint8_t *delta_share_out) {
    return lut_online(q, first_instance, batch, n_luts, n_in, arity, tt, lsrc, tcount, masked_in, lam_out, party_id,
                      delta_share_out, false);
}
// the same with the per-item arrays (masked_in, lam_out, delta_share_out: batch x n_luts bytes each) in DEVICE memory on
// the circuit's GPU: nothing crosses PCIe but the template tables the first time; queued on the context's stream and NOT
// waited for (sot_ctx_sync; the arrays must be complete when the call is made and stay valid until then)
extern "C" int sot_lut_online_eval_dev(sot_and_circ *q, uint32_t first_instance, uint32_t batch, uint32_t n_luts, uint32_t n_in,
                                       const uint32_t *arity, const uint32_t *tt, const uint32_t *lsrc, const uint32_t *tcount,
                                       const void *masked_in_dev, const void *lam_out_dev, uint32_t party_id,
                                       void *delta_share_dev) {
    return lut_online(q, first_instance, batch, n_luts, n_in, arity, tt, lsrc, tcount, (const uint8_t *)masked_in_dev,
                      (const uint8_t *)lam_out_dev, party_id, (uint8_t *)delta_share_dev, true);
}
