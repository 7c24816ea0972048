// libbusca_hip.so - C-ABI entry points (include/busca_hip.h) and host-side plumbing.
// gfx950 only.  Kernels live in the *.hip.inc files included below (one translation unit keeps the
// build a single hipcc invocation).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/busca_hip.h"

#include "dt_kernel.hip.inc"
#include "pairwise_kernel.hip.inc"
#include "crop_kernel.hip.inc"
#include "reid_kernel.hip.inc"

// ---------------------------------------------------------------------------------------------------------
struct DTState {
    bool loaded = false;
    busca_dt_cfg cfg{};
    void* dev_blob = nullptr;      // one allocation holding every packed matrix / vector / LUT
    size_t dev_bytes = 0;
    DTParams proto{};              // weight pointers filled in, per-call fields zero
};

struct busca_ctx {
    int device = 0;
    std::string err;
    DTState dt;
    ReidState reid;
    // kernel timing (HIP events on the launch stream)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pending;
    std::vector<hipEvent_t> ev_free;
    double t_ms = 0.0;
    long long t_n = 0;
    int* crop_fill = nullptr;      // per-crop pad value scratch (busca_crop_gather)
    int crop_fill_cap = 0;
};

static int fail(busca_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e__ = (call);                                                                           \
        if (e__ != hipSuccess) return fail((c), BUSCA_EHIP, "%s -> %s", #call, hipGetErrorString(e__));     \
    } while (0)

extern "C" int busca_version(void) { return 1000; }

static std::string g_create_err;   // busca_last_error(NULL) reports why busca_ctx_create failed

extern "C" int busca_ctx_create(int device, busca_ctx** out) {
    if (!out) return BUSCA_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_err = std::string("hipGetDeviceCount: ") + hipGetErrorString(e) + ", devices=" + std::to_string(n);
        return BUSCA_EHIP;
    }
    if (device < 0 || device >= n) { g_create_err = "device index out of range"; return BUSCA_EINVAL; }
    e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return BUSCA_EHIP; }
    busca_ctx* c = new busca_ctx();
    c->device = device;
    *out = c;
    return BUSCA_OK;
}

static void timing_drain(busca_ctx* c) {
    for (auto& pr : c->ev_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(pr.second) == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
            c->t_ms += ms;
            c->t_n += 1;
        }
        c->ev_free.push_back(pr.first);
        c->ev_free.push_back(pr.second);
    }
    c->ev_pending.clear();
}

extern "C" void busca_ctx_destroy(busca_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    timing_drain(c);
    for (auto e : c->ev_free) hipEventDestroy(e);
    if (c->dt.dev_blob) hipFree(c->dt.dev_blob);
    if (c->crop_fill) hipFree(c->crop_fill);
    reid_free(c->reid);
    delete c;
}

extern "C" const char* busca_last_error(const busca_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

extern "C" int busca_timing_enable(busca_ctx* c, int32_t on) {
    if (!c) return BUSCA_EINVAL;
    c->timing = on != 0;
    return BUSCA_OK;
}
extern "C" int busca_timing_read(busca_ctx* c, double* avg_ms, int64_t* launches, int32_t reset) {
    if (!c) return BUSCA_EINVAL;
    timing_drain(c);
    if (avg_ms) *avg_ms = c->t_n ? c->t_ms / (double)c->t_n : 0.0;
    if (launches) *launches = c->t_n;
    if (reset) { c->t_ms = 0.0; c->t_n = 0; }
    return BUSCA_OK;
}
static hipEvent_t timing_event(busca_ctx* c) {
    if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
struct TimedLaunch {   // RAII: records start/stop events around one kernel launch when timing is on
    busca_ctx* c; hipStream_t s; hipEvent_t e0{}, e1{}; bool on;
    TimedLaunch(busca_ctx* c_, hipStream_t s_) : c(c_), s(s_), on(c_->timing) {
        if (on) { e0 = timing_event(c); e1 = timing_event(c); hipEventRecord(e0, s); }
    }
    ~TimedLaunch() {
        if (on) { hipEventRecord(e1, s); c->ev_pending.emplace_back(e0, e1); if (c->ev_pending.size() > 4096) timing_drain(c); }
    }
};

// ---------------------------------------------------------------------------------------------------------
// Decision Transformer: weight blob -> device
// ---------------------------------------------------------------------------------------------------------
static bool dt_cfg_ok(const busca_dt_cfg* g) {
    if (!g) return false;
    if (!(g->d == 64 || g->d == 256 || g->d == 512)) return false;
    if (g->nhead != 4 || g->nlayers < 1 || g->nlayers > DT_MAX_LAYERS) return false;
    if (g->E != 512 || g->ff != 2 * g->d) return false;
    if (g->precision != BUSCA_PREC_F32 && g->precision != BUSCA_PREC_F16) return false;
    return true;
}

extern "C" size_t busca_dt_blob_floats(const busca_dt_cfg* g) {
    if (!dt_cfg_ok(g)) return 0;
    const size_t d = g->d, E = g->E, ff = g->ff;
    size_t n = d * E + d + 3 * d;
    n += (size_t)g->nlayers * (3 * d * d + 3 * d + d * d + d + ff * d + ff + d * ff + d + 4 * d);
    n += 3 * d + 1;
    return n;
}

// Pack W[N][K] (row-major f32) into MFMA operand-fragment order: for (row tile nt, chunk kc) 64 lanes x 16 B,
// lane (a = lane&15, kb = lane>>4) = W[16nt + a][kc*CHUNK + kb*SUB .. +SUB).  Returns bytes written.
static size_t pack_matrix(const float* W, int N, int K, int prec, unsigned char* dst) {
    const int chunk = prec == BUSCA_PREC_F32 ? 16 : 32, sub = chunk / 4;
    const int NT = N / 16, KC = K / chunk;
    size_t off = 0;
    for (int nt = 0; nt < NT; ++nt)
        for (int kc = 0; kc < KC; ++kc)
            for (int lane = 0; lane < 64; ++lane) {
                const int a = lane & 15, kb = lane >> 4;
                const float* src = W + (size_t)(16 * nt + a) * K + kc * chunk + kb * sub;
                if (prec == BUSCA_PREC_F32) {
                    memcpy(dst + off, src, 16);
                } else {
                    _Float16 h[8];
                    for (int i = 0; i < 8; ++i) h[i] = (_Float16)src[i];
                    memcpy(dst + off, h, 16);
                }
                off += 16;
            }
    return off;
}

extern "C" int busca_dt_load_weights(busca_ctx* c, const busca_dt_cfg* g, const float* blob, size_t blob_floats,
                                     const uint16_t* lut_xy, const uint16_t* lut_sz, const uint16_t* lut_t, int32_t lut_c) {
    if (!c) return BUSCA_EINVAL;
    if (!dt_cfg_ok(g)) return fail(c, BUSCA_EINVAL, "unsupported DT config (d in {64,256,512}, ff == 2d, nhead == 4, E == 512)");
    if (!blob || blob_floats != busca_dt_blob_floats(g)) return fail(c, BUSCA_EINVAL, "weight blob has %zu floats, expected %zu", blob_floats, busca_dt_blob_floats(g));
    if (!lut_xy || !lut_sz || !lut_t || lut_c <= 0 || 3 * lut_c < g->d) return fail(c, BUSCA_EINVAL, "bad encoding LUTs");
    HIP_TRY(c, hipSetDevice(c->device));
    const int d = g->d, E = g->E, ff = g->ff, prec = g->precision;
    const size_t es = prec == BUSCA_PREC_F32 ? 4 : 2;
    // host staging buffer, everything 256-byte aligned
    std::vector<unsigned char> host;
    auto reserve = [&](size_t bytes) { size_t off = (host.size() + 255) & ~(size_t)255; host.resize(off + bytes); return off; };
    const float* cur = blob;
    auto take = [&](size_t n) { const float* p = cur; cur += n; return p; };
    auto put_vec = [&](const float* src, size_t n) { size_t off = reserve(n * 4); memcpy(host.data() + off, src, n * 4); return off; };
    auto put_mat = [&](const float* src, int N, int K) { size_t off = reserve((size_t)N * K * es); pack_matrix(src, N, K, prec, host.data() + off); return off; };

    struct Offs { size_t w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2; } lo[DT_MAX_LAYERS];
    const size_t o_wemb = put_mat(take((size_t)d * E), d, E);
    const size_t o_bemb = put_vec(take(d), d);
    const size_t o_sep = put_vec(take(d), d), o_non = put_vec(take(d), d), o_bad = put_vec(take(d), d);
    for (int l = 0; l < g->nlayers; ++l) {
        lo[l].w_in = put_mat(take((size_t)3 * d * d), 3 * d, d);
        lo[l].b_in = put_vec(take(3 * d), 3 * d);
        lo[l].w_out = put_mat(take((size_t)d * d), d, d);
        lo[l].b_out = put_vec(take(d), d);
        lo[l].w1 = put_mat(take((size_t)ff * d), ff, d);
        lo[l].b1 = put_vec(take(ff), ff);
        lo[l].w2 = put_mat(take((size_t)d * ff), d, ff);
        lo[l].b2 = put_vec(take(d), d);
        lo[l].g1 = put_vec(take(d), d); lo[l].be1 = put_vec(take(d), d);
        lo[l].g2 = put_vec(take(d), d); lo[l].be2 = put_vec(take(d), d);
    }
    const size_t o_dg = put_vec(take(d), d), o_db = put_vec(take(d), d), o_dw = put_vec(take(d), d);
    const float dec_bias = *take(1);
    auto put_lut = [&](const uint16_t* src, size_t rows) { size_t off = reserve(rows * lut_c * 2); memcpy(host.data() + off, src, rows * lut_c * 2); return off; };
    const size_t o_lxy = put_lut(lut_xy, 211), o_lsz = put_lut(lut_sz, 211), o_lt = put_lut(lut_t, 61);

    DTState& S = c->dt;
    if (S.dev_blob) { HIP_TRY(c, hipDeviceSynchronize()); HIP_TRY(c, hipFree(S.dev_blob)); S.dev_blob = nullptr; S.loaded = false; }
    HIP_TRY(c, hipMalloc(&S.dev_blob, host.size()));
    HIP_TRY(c, hipMemcpy(S.dev_blob, host.data(), host.size(), hipMemcpyHostToDevice));
    S.dev_bytes = host.size();
    const char* base = (const char*)S.dev_blob;
    DTParams P{};
    P.w_embed = (const u32x4*)(base + o_wemb); P.b_embed = (const float*)(base + o_bemb);
    P.tok_sep = (const float*)(base + o_sep); P.tok_non = (const float*)(base + o_non); P.tok_bad = (const float*)(base + o_bad);
    for (int l = 0; l < g->nlayers; ++l) {
        DTLayerW& W = P.layer[l];
        W.w_in = (const u32x4*)(base + lo[l].w_in); W.b_in = (const float*)(base + lo[l].b_in);
        W.w_out = (const u32x4*)(base + lo[l].w_out); W.b_out = (const float*)(base + lo[l].b_out);
        W.w1 = (const u32x4*)(base + lo[l].w1); W.b1 = (const float*)(base + lo[l].b1);
        W.w2 = (const u32x4*)(base + lo[l].w2); W.b2 = (const float*)(base + lo[l].b2);
        W.g1 = (const float*)(base + lo[l].g1); W.be1 = (const float*)(base + lo[l].be1);
        W.g2 = (const float*)(base + lo[l].g2); W.be2 = (const float*)(base + lo[l].be2);
    }
    P.dec_g = (const float*)(base + o_dg); P.dec_b = (const float*)(base + o_db); P.dec_w = (const float*)(base + o_dw);
    P.dec_bias = dec_bias;
    P.lut_xy = (const _Float16*)(base + o_lxy); P.lut_sz = (const _Float16*)(base + o_lsz); P.lut_t = (const _Float16*)(base + o_lt);
    P.lut_c = lut_c;
    P.nlayers = g->nlayers; P.act = g->activation; P.fake_f64 = g->fake_bbox_f64;
    S.proto = P;
    S.cfg = *g;
    S.loaded = true;
    return BUSCA_OK;
}

// ---------------------------------------------------------------------------------------------------------
template <int PREC, int MT, int D, int FF, int NCH>
static int dt_launch(busca_ctx* c, const DTParams& P, hipStream_t s) {
    typedef DTLds<PREC, MT, D, FF, 512, NCH> LD;
    static_assert(LD::TOTAL <= 160 * 1024, "LDS plan exceeds 160 KiB");
    auto kern = dt_fused_kernel<PREC, MT, D, FF, 512, NCH>;
    static bool attr_set = false;   // per instantiation
    if (!attr_set) {
        HIP_TRY(c, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LD::TOTAL));
        attr_set = true;
    }
    static const bool prof = getenv("BUSCA_DT_PROF") != nullptr;   // debug: phase timestamps of workgroup 0
    if (prof) {
        DTParams Q = P;
        long long* d = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d, 4 * DT_PROF_SLOTS * sizeof(long long)));
        HIP_TRY(c, hipMemset(d, 0, 4 * DT_PROF_SLOTS * sizeof(long long)));
        Q.prof = d;
        hipLaunchKernelGGL(kern, dim3(P.B), dim3(256), LD::TOTAL, s, Q);
        HIP_TRY(c, hipStreamSynchronize(s));
        long long h[4 * DT_PROF_SLOTS];
        HIP_TRY(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipFree(d));
        fprintf(stderr, "DT_PROF grid=%d:", P.B);
        for (int w = 0; w < 4; ++w) {
            fprintf(stderr, "\n w%d", w);
            for (int i = 1; i < DT_PROF_SLOTS; ++i)
                if (h[w * DT_PROF_SLOTS + i]) fprintf(stderr, " %d:%lld", i, h[w * DT_PROF_SLOTS + i] - h[w * DT_PROF_SLOTS]);
        }
        fprintf(stderr, "\n");
        return BUSCA_OK;
    }
    {
        TimedLaunch tl(c, s);
        hipLaunchKernelGGL(kern, dim3(P.B), dim3(256), LD::TOTAL, s, P);
    }
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

extern "C" int busca_dt_forward(busca_ctx* c, const float* mem_feat, const float* can_feat, const float* mem_ltrb,
                                const float* can_ltrb, int32_t B, int32_t L, int32_t P, float* logits, float* probs,
                                int32_t* argmax, float* hidden, float* att, void* stream) {
    if (!c) return BUSCA_EINVAL;
    if (!c->dt.loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_dt_forward before busca_dt_load_weights");
    if (B < 0 || L < 1 || P < 1 || !logits) return fail(c, BUSCA_EINVAL, "bad shape B=%d L=%d P=%d or null logits", B, L, P);
    if (B == 0) return BUSCA_OK;
    if (!mem_feat || !can_feat || !mem_ltrb || !can_ltrb) return fail(c, BUSCA_EINVAL, "null input pointer");
    if (P + 2 > 64) return fail(c, BUSCA_EINVAL, "P=%d: the fused path supports at most 62 proposals per track", P);
    DTParams K = c->dt.proto;
    K.mem_feat = mem_feat; K.can_feat = can_feat; K.mem_ltrb = mem_ltrb; K.can_ltrb = can_ltrb;
    K.logits = logits; K.probs = probs; K.argmax = argmax; K.hidden = hidden; K.att = att;
    K.B = B; K.L = L; K.P = P; K.T = L + 2 * (P + 2);
    const int MT = (K.T + 15) / 16;
    const int d = c->dt.cfg.d, prec = c->dt.cfg.precision;
    hipStream_t s = (hipStream_t)stream;
#define DT_CASE(PR, M, DD, NCH) if (prec == PR && MT == M && d == DD) return dt_launch<PR, M, DD, 2 * DD, NCH>(c, K, s)
    DT_CASE(0, 1, 64, 1); DT_CASE(0, 1, 256, 1); DT_CASE(0, 1, 512, 1); DT_CASE(1, 1, 64, 1); DT_CASE(1, 1, 256, 1); DT_CASE(1, 1, 512, 1);
    DT_CASE(0, 2, 64, 1); DT_CASE(0, 3, 64, 1); DT_CASE(0, 4, 64, 1);
    DT_CASE(0, 2, 256, 1); DT_CASE(0, 3, 256, 1);
    DT_CASE(0, 2, 512, 2);
    DT_CASE(1, 2, 64, 1); DT_CASE(1, 3, 64, 1); DT_CASE(1, 4, 64, 1);
    DT_CASE(1, 2, 256, 1); DT_CASE(1, 3, 256, 1); DT_CASE(1, 4, 256, 1); DT_CASE(1, 5, 256, 1);
    DT_CASE(1, 2, 512, 1); DT_CASE(1, 3, 512, 1); DT_CASE(1, 4, 512, 2);
#undef DT_CASE
    return fail(c, BUSCA_EINVAL, "no fused DT kernel for T=%d (tiles %d), d=%d, precision=%d", K.T, MT, d, prec);
}

extern "C" int busca_dt_bucket_ids(busca_ctx* c, const float* mem_ltrb, const float* can_ltrb, int32_t B, int32_t L,
                                   int32_t P, int32_t* ids, void* stream) {
    if (!c || !ids || B < 0 || L < 1 || P < 1) return fail(c, BUSCA_EINVAL, "bad arguments");
    if (B == 0) return BUSCA_OK;
    const int fake64 = c->dt.loaded ? c->dt.cfg.fake_bbox_f64 : 1;
    const int n = B * (L + 2 * (P + 2));
    hipLaunchKernelGGL(dt_bucket_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mem_ltrb, can_ltrb, B, L, P, fake64, ids);
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

#include "capi_geometry.hip.inc"
#include "capi_reid.hip.inc"
