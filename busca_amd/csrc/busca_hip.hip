// libbusca_hip.so, core unit - C-ABI of include/busca_hip.h over the HIP kernels in this directory (the other units: busca_internal.hpp).
// Written for gfx950 (MI355X) only.  hipcc --offload-arch=gfx950 (see busca_amd/build.py).
#include "busca_internal.hpp"

#define BUSCA_DT_BUCKET_KERNEL 1
#include "dt_kernel.hip.inc"
#include "pairwise_kernel.hip.inc"
#include "track_kernel.hip.inc"
#include "crop_kernel.hip.inc"
#include "ecc_kernel.hip.inc"

int ensure_lds(busca_ctx* c, const void* kern, size_t bytes) {
    if (c->lds_configured.count(kern)) return BUSCA_OK;
    HIP_TRY(c, hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    c->lds_configured.insert(kern);
    return BUSCA_OK;
}

extern "C" int busca_version(void) { return 2000; }

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
    c->opt.from_env();
    c->reid = reid_state_new();
    *out = c;
    return BUSCA_OK;
}

#ifndef BUSCA_BUILD_FLAGS
#define BUSCA_BUILD_FLAGS "unknown"
#endif
extern "C" const char* busca_build_info(void) { return "libbusca_hip gfx950 flags: " BUSCA_BUILD_FLAGS; }

extern "C" int busca_set_option(busca_ctx* c, const char* name, int32_t value) {
    if (!c || !name) return BUSCA_EINVAL;
    BuscaOptions& o = c->opt;
    const std::string n(name);
    if (n.rfind("reid_", 0) == 0) return reid_set_option(c, name, value);
    if (n == "dt_ntrk") o.dt_ntrk = value;
    else if (n == "dt_split") o.dt_split = value;
    else if (n == "dt_tiled") o.dt_tiled = value;
    else if (n == "crop_band") o.crop_band = value;
    else if (n == "dt_exact_f32") o.dt_exact_f32 = value != 0;
    else if (n == "dt_status") { if (c->dt.xerr) *c->dt.xerr = value; }      // 0 = the caller has read the status of its synchronised forward and dealt with it
    else return fail(c, BUSCA_EINVAL, "busca_set_option: unknown option '%s'", name);
    return BUSCA_OK;
}
extern "C" int busca_get_option(busca_ctx* c, const char* name, int32_t* value) {
    if (!c || !name || !value) return BUSCA_EINVAL;
    const BuscaOptions& o = c->opt;
    const std::string n(name);
    if (n.rfind("reid_", 0) == 0) return reid_get_option(c, name, value);
    if (n == "dt_ntrk") *value = o.dt_ntrk;
    else if (n == "dt_split") *value = o.dt_split;
    else if (n == "last_dt_split") *value = o.last_dt_split;
    else if (n == "dt_status") *value = c->dt.xerr ? *c->dt.xerr : 0;       // 0 ok, 1 a split launch lost a partner, 2 an x3 forward clipped an operand; valid once the forward's stream is
                                                                            // synchronised; cleared by busca_set_option("dt_status", 0) (an uncleared status is also returned by the next forward)
    else if (n == "dt_tiled") *value = o.dt_tiled;
    else if (n == "crop_band") *value = o.crop_band;
    else if (n == "dt_exact_f32") *value = o.dt_exact_f32;
    else if (n == "last_dt_grid") *value = o.last_dt_grid;
    else if (n == "last_dt_ntrk") *value = o.last_dt_ntrk;
    else return fail(c, BUSCA_EINVAL, "busca_get_option: unknown option '%s'", name);
    return BUSCA_OK;
}

void timing_drain(busca_ctx* c) {
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
    if (c->dt.dev_tiled) hipFree(c->dt.dev_tiled);
    if (c->dt.ws) hipFree(c->dt.ws);
    if (c->dt.dev_blob32) hipFree(c->dt.dev_blob32);
    if (c->dt.xch) hipFree(c->dt.xch);
    if (c->dt.xflag) hipFree(c->dt.xflag);
    if (c->dt.xlg) hipFree(c->dt.xlg);
    if (c->dt.xerr) hipHostFree(c->dt.xerr);
    if (c->crop_fill) hipFree(c->crop_fill);
    if (c->ecc_ws) hipFree(c->ecc_ws);
    reid_state_delete(c->reid);
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
hipEvent_t timing_event(busca_ctx* c) {
    if (!c->ev_free.empty()) { hipEvent_t e = c->ev_free.back(); c->ev_free.pop_back(); return e; }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
// ---------------------------------------------------------------------------------------------------------
// Decision Transformer: weight blob -> device
// ---------------------------------------------------------------------------------------------------------
static bool dt_cfg_ok(const busca_dt_cfg* g) {
    if (!g) return false;
    if (!(g->d == 64 || g->d == 256 || g->d == 512)) return false;
    if (g->nlayers < 1 || g->nlayers > DT_MAX_LAYERS) return false;
    if (g->nhead < 1 || g->d % g->nhead != 0) return false;
    const int hd = g->d / g->nhead;
    if (!(hd == 16 || hd == 32 || hd == 64 || hd == 128)) return false;             // head widths the attention kernels are built for
    if (g->E != 512 || g->ff < g->d || g->ff % g->d != 0 || g->ff > 8 * g->d) return false;   // ff = k d: whole column blocks of the layer-wise GEMM
    if (g->precision != BUSCA_PREC_F32 && g->precision != BUSCA_PREC_F16 && g->precision != BUSCA_PREC_F16X3) return false;
    if (g->layout & ~(BUSCA_LAYOUT_CAN_FIRST | BUSCA_LAYOUT_NO_BAD | BUSCA_LAYOUT_SEP_AS_CAN)) return false;
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
// BUSCA_PREC_F16X3: per (nt, kc) the 64 hi fragments, then the 64 lo fragments, of DT_X3_WS * W split as hi = fp16(x), lo = fp16(x - hi).
static size_t pack_matrix(const float* W, int N, int K, int prec, unsigned char* dst) {
    if (prec == BUSCA_PREC_F16X3) {
        const int NT = N / 16, KC = K / 32;
        size_t off = 0;
        for (int nt = 0; nt < NT; ++nt)
            for (int kc = 0; kc < KC; ++kc) {
                for (int lane = 0; lane < 64; ++lane) {
                    const int a = lane & 15, kb = lane >> 4;
                    const float* src = W + (size_t)(16 * nt + a) * K + kc * 32 + kb * 8;
                    _Float16 h[8], l[8];
                    for (int i = 0; i < 8; ++i) {
                        float x = src[i] * DT_X3_WS;
                        x = x > 65504.f ? 65504.f : (x < -65504.f ? -65504.f : x);
                        h[i] = (_Float16)x; l[i] = (_Float16)(x - (float)h[i]);
                    }
                    memcpy(dst + off + (size_t)lane * 16, h, 16);
                    memcpy(dst + off + 1024 + (size_t)lane * 16, l, 16);
                }
                off += 2048;
            }
        return off;
    }
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
    if (!dt_cfg_ok(g)) return fail(c, BUSCA_EINVAL, "unsupported DT config (d in {64,256,512}, d / nhead in {16,32,64,128}, ff = k d with k <= 8, E == 512)");
    if (!blob || blob_floats != busca_dt_blob_floats(g)) return fail(c, BUSCA_EINVAL, "weight blob has %zu floats, expected %zu", blob_floats, busca_dt_blob_floats(g));
    if (!lut_xy || !lut_sz || !lut_t || lut_c <= 0 || 3 * lut_c < g->d) return fail(c, BUSCA_EINVAL, "bad encoding LUTs");
    HIP_TRY(c, hipSetDevice(c->device));
    const int d = g->d, E = g->E, ff = g->ff, prec = g->precision;
    const size_t es = prec == BUSCA_PREC_F16 ? 2 : 4;      // (x3: two fp16 planes)
    // host staging buffer, everything 256-byte aligned
    std::vector<unsigned char> host;
    auto reserve = [&](size_t bytes) { size_t off = (host.size() + 255) & ~(size_t)255; host.resize(off + bytes); return off; };
    const float* cur = blob;
    auto take = [&](size_t n) { const float* p = cur; cur += n; return p; };
    auto put_vec = [&](const float* src, size_t n) { size_t off = reserve(n * 4); memcpy(host.data() + off, src, n * 4); return off; };
    float wmax = 0.f;          // largest matrix entry (x3: the split-fp16 flavour carries DT_X3_WS * w in fp16 hi + lo - a checkpoint beyond that range must not be clipped silently)
    auto put_mat = [&](const float* src, int N, int K) {
        size_t off = reserve((size_t)N * K * es);
        for (size_t i = 0; i < (size_t)N * K; ++i) wmax = std::max(wmax, std::fabs(src[i]));
        pack_matrix(src, N, K, prec, host.data() + off);
        return off;
    };

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
    double dec_cb = (double)dec_bias;                    // (the three decoder vectors sit right in front of the bias)
    for (int f = 0; f < d; ++f) dec_cb += (double)(cur - 1 - 2 * d)[f] * (double)(cur - 1 - d)[f];
    auto put_lut = [&](const uint16_t* src, size_t rows) { size_t off = reserve(rows * lut_c * 2); memcpy(host.data() + off, src, rows * lut_c * 2); return off; };
    const size_t o_lxy = put_lut(lut_xy, 211), o_lsz = put_lut(lut_sz, 211), o_lt = put_lut(lut_t, 61);

    if (prec == BUSCA_PREC_F16X3 && !(wmax * DT_X3_WS <= 65504.f))
        return fail(c, BUSCA_EINVAL, "Decision-Transformer weights reach |w| = %g, beyond the split-fp16 (BUSCA_PREC_F16X3) operand range of %g: load them with BUSCA_PREC_F32", (double)wmax, 65504.0 / DT_X3_WS);
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
    P.dec_bias = dec_bias; P.dec_cb = (float)dec_cb;
    P.lut_xy = (const _Float16*)(base + o_lxy); P.lut_sz = (const _Float16*)(base + o_lsz); P.lut_t = (const _Float16*)(base + o_lt);
    P.lut_c = lut_c;
    P.nlayers = g->nlayers; P.act = g->activation; P.fake_f64 = g->fake_bbox_f64;
    P.can_pos = (g->layout & BUSCA_LAYOUT_CAN_FIRST) ? 0 : 1; P.nspec = (g->layout & BUSCA_LAYOUT_NO_BAD) ? 1 : 2;
    P.sep_can = (g->layout & BUSCA_LAYOUT_SEP_AS_CAN) ? 1 : 0;
    if (P.nspec == 1) P.fake_f64 = 0;    // the float64 promotion comes from torch.cat with the float64 BAD box (encodings.py:127-140): no BAD, no promotion
    S.proto = P;
    S.proto32 = P;
    if (S.dev_blob32) { HIP_TRY(c, hipFree(S.dev_blob32)); S.dev_blob32 = nullptr; }
    if (prec == BUSCA_PREC_F16X3) {
        // the layer-wise kernels that take fragment-packed weights (dtl_qkv_attn_kernel, dtl_ffn_kernel) run in exact f32 for this flavour: a second packing
        std::vector<unsigned char> h32;
        auto put32 = [&](const float* src, int N, int K) { size_t off = (h32.size() + 255) & ~(size_t)255; h32.resize(off + (size_t)N * K * 4); pack_matrix(src, N, K, BUSCA_PREC_F32, h32.data() + off); return off; };
        const float* q = blob;
        const size_t e32 = put32(q, d, E); q += (size_t)d * E + d + 3 * d;
        size_t oi[DT_MAX_LAYERS], oo[DT_MAX_LAYERS], o1[DT_MAX_LAYERS], o2[DT_MAX_LAYERS];
        for (int l = 0; l < g->nlayers; ++l) {
            oi[l] = put32(q, 3 * d, d); q += (size_t)3 * d * d + 3 * d;
            oo[l] = put32(q, d, d); q += (size_t)d * d + d;
            o1[l] = put32(q, ff, d); q += (size_t)ff * d + ff;
            o2[l] = put32(q, d, ff); q += (size_t)d * ff + d + 4 * d;
        }
        HIP_TRY(c, hipMalloc(&S.dev_blob32, h32.size()));
        HIP_TRY(c, hipMemcpy(S.dev_blob32, h32.data(), h32.size(), hipMemcpyHostToDevice));
        const char* b32 = (const char*)S.dev_blob32;
        S.proto32.w_embed = (const u32x4*)(b32 + e32);
        for (int l = 0; l < g->nlayers; ++l) {
            DTLayerW& W = S.proto32.layer[l];
            W.w_in = (const u32x4*)(b32 + oi[l]); W.w_out = (const u32x4*)(b32 + oo[l]); W.w1 = (const u32x4*)(b32 + o1[l]); W.w2 = (const u32x4*)(b32 + o2[l]);
        }
    }
    S.cfg = *g;
    if (S.dev_tiled) { HIP_TRY(c, hipFree(S.dev_tiled)); S.dev_tiled = nullptr; }
    S.tw = DTTiledW();
    {   // plain row-major matrices in the operand type for the tiled (layer-wise) path
        std::vector<unsigned char> hw;
        auto putm = [&](const float* src, size_t n) {
            while (hw.size() % 16) hw.push_back(0);
            const size_t off = hw.size();
            hw.resize(off + n * es);
            if (prec != BUSCA_PREC_F16) memcpy(hw.data() + off, src, n * 4);      // (x3: shapes beyond the fused kernel run the exact f32 layer-wise path)
            else { _Float16* d16 = (_Float16*)(hw.data() + off); for (size_t i = 0; i < n; ++i) d16[i] = (_Float16)src[i]; }
            return off;
        };
        const float* q = blob;
        const size_t o_e = putm(q, (size_t)d * E); q += (size_t)d * E + d + 3 * d;
        size_t oi[DT_MAX_LAYERS], oo[DT_MAX_LAYERS], o1[DT_MAX_LAYERS], o2[DT_MAX_LAYERS];
        for (int l = 0; l < g->nlayers; ++l) {
            oi[l] = putm(q, (size_t)3 * d * d); q += (size_t)3 * d * d + 3 * d;
            oo[l] = putm(q, (size_t)d * d); q += (size_t)d * d + d;
            o1[l] = putm(q, (size_t)ff * d); q += (size_t)ff * d + ff;
            o2[l] = putm(q, (size_t)d * ff); q += (size_t)d * ff + d + 4 * d;
        }
        HIP_TRY(c, hipMalloc(&S.dev_tiled, hw.size()));
        HIP_TRY(c, hipMemcpy(S.dev_tiled, hw.data(), hw.size(), hipMemcpyHostToDevice));
        const char* tb = (const char*)S.dev_tiled;
        S.tw.w_embed = tb + o_e;
        for (int l = 0; l < g->nlayers; ++l) { S.tw.w_in[l] = tb + oi[l]; S.tw.w_out[l] = tb + oo[l]; S.tw.w1[l] = tb + o1[l]; S.tw.w2[l] = tb + o2[l]; }
    }
    {   // token-split tail: its exchange buffers (64 / 128 MiB at d = 256 / 512) are allocated by the first launch that splits (dt_split_ensure) - a context that
        // never splits (the f16 flavour by default, bench.py's extra contexts) never pays for them; only the status word lives from here on
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) S.num_cu = prop.multiProcessorCount; }
        if (S.xch) { HIP_TRY(c, hipDeviceSynchronize()); HIP_TRY(c, hipFree(S.xch)); S.xch = nullptr; }
        if (S.xflag) { HIP_TRY(c, hipFree(S.xflag)); S.xflag = nullptr; }
        if (S.xlg) { HIP_TRY(c, hipFree(S.xlg)); S.xlg = nullptr; }
        S.xslots = d % 64 == 0 ? S.num_cu : 0;
        if (!S.xerr) {
            HIP_TRY(c, hipHostMalloc((void**)&S.xerr, sizeof(int), hipHostMallocMapped)); *S.xerr = 0;
            HIP_TRY(c, hipHostGetDevicePointer((void**)&S.xerr_dev, S.xerr, 0));
        }
        S.xepoch = 0;
    }
    S.loaded = true;
    return BUSCA_OK;
}

// Workspace of the layer-wise path for M rows (bytes).  Grown outside the forward by busca_dt_reserve; a forward that finds
// it too small grows it itself (one stream synchronisation + hipMalloc, first call of a larger shape only).
size_t dtl_ws_bytes(size_t M, int D, int FF, size_t es) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(M * D * 4) + al(M * D * 2) + al(M * 3 * D * es) + al(M * D * es) + al(M * FF * es) + al(M * 3 * 4) + al(M * 4);
}

int dtl_ws_ensure(busca_ctx* c, size_t need, hipStream_t s) {
    DTState& S = c->dt;
    if (S.ws_bytes >= need) return BUSCA_OK;
    if (S.ws) { HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(S.ws)); S.ws = nullptr; S.ws_bytes = 0; }
    if (hipMalloc(&S.ws, need) != hipSuccess) return fail(c, BUSCA_ENOMEM, "cannot allocate %zu bytes of DT workspace", need);
    S.ws_bytes = need;
    return BUSCA_OK;
}

// Exchange buffers of the token-split tail: one round's worth of tracks (two workgroups each), sized for the loaded width; allocated by the first launch that
// splits (one device synchronisation + three hipMallocs, once per weight set).
int dt_split_ensure(busca_ctx* c) {
    DTState& S = c->dt;
    if (S.xch != nullptr) return BUSCA_OK;
    if (S.xslots <= 0) return fail(c, BUSCA_EINVAL, "token-split launch without exchange slots");
    const int d = S.cfg.d;
    const size_t per_slot = (size_t)2 * DT_XMAX_MT * 4 * (2 * (d / 64)) * 1024;       // parity x tile x wave x (K, V tiles of a head) x 64 lanes x 16 bytes
    HIP_TRY(c, hipMalloc(&S.xch, per_slot * S.xslots));
    HIP_TRY(c, hipMalloc((void**)&S.xflag, (size_t)S.xslots * DT_XMAX_MT * DT_XFLAGS * sizeof(unsigned)));
    HIP_TRY(c, hipMemset(S.xflag, 0, (size_t)S.xslots * DT_XMAX_MT * DT_XFLAGS * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc((void**)&S.xlg, (size_t)S.xslots * DT_XMAX_MT * 16 * sizeof(float)));
    S.xepoch = 0;
    HIP_TRY(c, hipDeviceSynchronize());
    return BUSCA_OK;
}

void dt_bucket_ids_launch(busca_ctx* c, hipStream_t s, const float* mem_ltrb, const float* can_ltrb, int B, int L, int P, int fake_f64, int can_pos, int nspec, int sep_can, int* ids) {
    const size_t n = (size_t)B * (L + 2 * (P + nspec));
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(dt_bucket_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, mem_ltrb, can_ltrb, B, L, P, fake_f64, can_pos, nspec, sep_can, ids);
}

extern "C" int busca_dt_reserve(busca_ctx* c, int32_t B, int32_t L, int32_t P, void* stream) {
    if (!c) return BUSCA_EINVAL;
    if (!c->dt.loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_dt_reserve before busca_dt_load_weights");
    if (B < 0 || L < 1 || P < 1) return fail(c, BUSCA_EINVAL, "bad shape B=%d L=%d P=%d", B, L, P);
    const size_t es = c->dt.cfg.precision == BUSCA_PREC_F16 ? 2 : 4;
    return dtl_ws_ensure(c, dtl_ws_bytes((size_t)B * (L + 2 * (P + c->dt.proto.nspec)), c->dt.cfg.d, c->dt.cfg.ff, es), (hipStream_t)stream);
}

static int dt_forward_impl(busca_ctx* c, const float* mem_feat, const float* can_feat, const float* mem_ltrb,
                           const float* can_ltrb, int32_t B, int32_t L, int32_t P, float* logits, float* probs,
                           int32_t* argmax, float* hidden, float* att, void* stream) {
    if (!c->dt.loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_dt_forward before busca_dt_load_weights");
    if (B < 0 || L < 1 || P < 1 || !logits) return fail(c, BUSCA_EINVAL, "bad shape B=%d L=%d P=%d or null logits", B, L, P);
    if (B == 0) return BUSCA_OK;
    if (!mem_feat || !can_feat || !mem_ltrb || !can_ltrb) return fail(c, BUSCA_EINVAL, "null input pointer");
    DTParams K = c->dt.proto;
    K.mem_feat = mem_feat; K.can_feat = can_feat; K.mem_ltrb = mem_ltrb; K.can_ltrb = can_ltrb;
    K.logits = logits; K.probs = probs; K.argmax = argmax; K.hidden = hidden; K.att = att;
    K.B = B; K.L = L; K.P = P; K.T = L + 2 * (P + K.nspec);
    K.xerr = c->dt.xerr_dev;
    const int MT = (K.T + 15) / 16;
    const int d = c->dt.cfg.d;
    int prec = c->dt.cfg.precision;
    if (prec == BUSCA_PREC_F16X3 && c->opt.dt_exact_f32) {      // exact float32 on the f32 fragment packing of the same matrices (how the host re-runs a clipped x3 step)
        prec = BUSCA_PREC_F32;
        K.w_embed = c->dt.proto32.w_embed;
        for (int l = 0; l < K.nlayers; ++l) { K.layer[l].w_in = c->dt.proto32.layer[l].w_in; K.layer[l].w_out = c->dt.proto32.layer[l].w_out; K.layer[l].w1 = c->dt.proto32.layer[l].w1; K.layer[l].w2 = c->dt.proto32.layer[l].w2; }
    }
    hipStream_t s = (hipStream_t)stream;
    const bool force_tiled = c->opt.dt_tiled != 0;          // testing: run the layer-wise path on any shape
    // the one-kernel path is built for the shipped geometry (four heads, ff = 2 d); other head counts / widths run layer-wise
    const bool fused_ok = !force_tiled && P + K.nspec <= 64 && c->dt.cfg.nhead == 4 && c->dt.cfg.ff == 2 * c->dt.cfg.d;
    // f16, d = 256: from two rounds of workgroups on (B > 256 CUs) a workgroup can take TWO tracks (each streamed weight fragment feeds
    // twice the tokens).  A two-track workgroup takes 1.88x as long as a one-track one (0.177 vs 0.094 ms per round, round 3), so the
    // flavour with the shorter sum of rounds is taken: two tracks at 257-512 and 769-1024 tracks, one track at 513-768 (three rounds of
    // 0.094 against two of 0.177) ... (BUSCA_DT_NTRK / option "dt_ntrk" = 1 / 2 forces either)
    const int ntrk_env = c->opt.dt_ntrk;
    const long r1 = (B + 255) / 256, r2 = (B + 511) / 512;
    const bool two = ntrk_env == 2 || (ntrk_env == 0 && B > 256 && r2 * 188 <= r1 * 100);
    if (fused_ok) {
        int rc = BUSCA_ENOKERNEL;
        if (prec == BUSCA_PREC_F32) rc = dt_fused_f32(c, K, MT, d, s);
        else if (prec == BUSCA_PREC_F16) rc = dt_fused_f16(c, K, MT, d, two, s);
        else rc = dt_fused_x3(c, K, MT, d, s);
        if (rc != BUSCA_ENOKERNEL) return rc;
    }
    // shapes beyond the fused kernel's on-chip plan: layer-wise tiled path, same arithmetic type (x3: the exact f32 layer-wise path on the f32 packing, proto32)
    if (prec == BUSCA_PREC_F16) return dt_tiled_f16(c, K, d, s);
    if (prec == BUSCA_PREC_F16X3) {
        // the split-fp16 layer kernels where they are built (d >= 256, four heads, ff a multiple of d; T <= 80 for the fused QKV + attention kernel - beyond it
        // that half of a layer runs the exact f32 kernels); else the exact f32 layer-wise path on the f32 packing (proto32)
        if (d >= 256 && c->dt.cfg.nhead == 4) {
            const int rc = dt_tiled_x3(c, K, d, s);
            if (rc != BUSCA_ENOKERNEL) return rc;
        }
        K.w_embed = c->dt.proto32.w_embed;
        for (int l = 0; l < K.nlayers; ++l) { K.layer[l].w_in = c->dt.proto32.layer[l].w_in; K.layer[l].w_out = c->dt.proto32.layer[l].w_out; K.layer[l].w1 = c->dt.proto32.layer[l].w1; K.layer[l].w2 = c->dt.proto32.layer[l].w2; }
    }
    return dt_tiled_f32(c, K, d, s);
}

extern "C" int busca_dt_forward(busca_ctx* c, const float* mem_feat, const float* can_feat, const float* mem_ltrb,
                                const float* can_ltrb, int32_t B, int32_t L, int32_t P, float* logits, float* probs,
                                int32_t* argmax, float* hidden, float* att, void* stream) {
    if (!c) return BUSCA_EINVAL;
    // Backstop for callers that never read "dt_status" after synchronising (busca_amd's own wrappers do, and re-run a clipped step in exact f32): a status a
    // kernel of an EARLIER call left in host-mapped memory is taken here, THIS call's kernels are launched all the same, and the earlier failure is returned.
    int st = 0;
    if (c->dt.xerr && *c->dt.xerr) { st = *c->dt.xerr; *c->dt.xerr = 0; }
    const int rc = dt_forward_impl(c, mem_feat, can_feat, mem_ltrb, can_ltrb, B, L, P, logits, probs, argmax, hidden, att, stream);
    if (rc != BUSCA_OK || st == 0) return rc;
    if (st == 2) return fail(c, BUSCA_EINVAL, "an EARLIER Decision-Transformer forward clipped activations beyond the split-fp16 (BUSCA_PREC_F16X3) operand range (|x| > 1023.5): "
                                              "its results were not float32-equivalent (re-run it with busca_set_option(\"dt_exact_f32\", 1) or load the model with BUSCA_PREC_F32); "
                                              "this call's kernels were launched normally");
    return fail(c, BUSCA_EHIP, "an EARLIER token-split Decision-Transformer launch gave up waiting for a partner workgroup: its results were invalid; this call's kernels were launched normally");
}

extern "C" int busca_dt_bucket_ids(busca_ctx* c, const float* mem_ltrb, const float* can_ltrb, int32_t B, int32_t L,
                                   int32_t P, int32_t* ids, void* stream) {
    if (!c || !ids || B < 0 || L < 1 || P < 1) return fail(c, BUSCA_EINVAL, "bad arguments");
    if (B == 0) return BUSCA_OK;
    const int fake64 = c->dt.loaded ? c->dt.proto.fake_f64 : 1;
    const int can_pos = c->dt.loaded ? c->dt.proto.can_pos : 1, nspec = c->dt.loaded ? c->dt.proto.nspec : 2, sep_can = c->dt.loaded ? c->dt.proto.sep_can : 0;
    dt_bucket_ids_launch(c, (hipStream_t)stream, mem_ltrb, can_ltrb, B, L, P, fake64, can_pos, nspec, sep_can, ids);
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

#include "capi_geometry.hip.inc"
