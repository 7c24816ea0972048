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
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/busca_hip.h"

#include "dt_kernel.hip.inc"
#include "pairwise_kernel.hip.inc"
#include "track_kernel.hip.inc"
#include "crop_kernel.hip.inc"
#include "reid_kernel.hip.inc"
#include "reid_gram.hip.inc"
#include "reid_halo.hip.inc"
#include "reid_kwave.hip.inc"
#include "reid_pipe.hip.inc"
#include "reid_f32.hip.inc"
#include "reid_x3.hip.inc"
#include "reid_x3p.hip.inc"
#include "dt_tiled.hip.inc"
#include "ecc_kernel.hip.inc"

// ---------------------------------------------------------------------------------------------------------
struct DTTiledW {                   // row-major copies of the matrices for the tiled path, in the operand type (f16 or f32)
    const void* w_embed = nullptr;
    const void *w_in[DT_MAX_LAYERS] = {}, *w_out[DT_MAX_LAYERS] = {}, *w1[DT_MAX_LAYERS] = {}, *w2[DT_MAX_LAYERS] = {};
};

struct DTState {
    bool loaded = false;
    busca_dt_cfg cfg{};
    void* dev_blob = nullptr;      // one allocation holding every packed matrix / vector / LUT
    size_t dev_bytes = 0;
    DTParams proto{};              // weight pointers filled in, per-call fields zero
    void* dev_blob32 = nullptr;    // x3 only: the matrices once more, packed for the exact f32 kernels (the layer-wise path of shapes beyond the one-kernel path)
    DTParams proto32{};            // = proto with the matrix pointers into dev_blob32
    void* dev_tiled = nullptr;     // row-major f16 matrices (tiled path)
    DTTiledW tw;
    void* ws = nullptr; size_t ws_bytes = 0;   // tiled-path activation workspace
    // token-split tail of the fused kernel (dt_fused_kernel<..., SPLIT = true>): K / V exchange tiles, flags, decoder hand-over of up to xslots tracks; xepoch
    // numbers the launches (flags only ever grow, nothing is cleared between launches); xerr is host memory the kernel writes if a wait ran out
    void* xch = nullptr; unsigned* xflag = nullptr; float* xlg = nullptr; int* xerr = nullptr; int* xerr_dev = nullptr; int xslots = 0; unsigned xepoch = 0;
    int num_cu = 256;
};

// Developer options of one context.  Defaults come from the environment ONCE, when the context is created; afterwards they
// change only through busca_set_option (so a test can flip a flavour between two forwards, and no forward calls getenv).
struct BuscaOptions {
    int dt_ntrk = 0;          // BUSCA_DT_NTRK: tracks per workgroup of the f16 fused kernel (0 = automatic: 2 from B > 256, d = 256)
    int dt_tiled = 0;         // BUSCA_DT_TILED: force the layer-wise Decision-Transformer path
    int dtl_rt = 0;           // BUSCA_DTL_RT: 2 / 4 = 64- / 128-row tiles of the layer-wise GEMMs (0 = automatic)
    int dtl_rt_mask = -1;     // BUSCA_DTL_RT_MASK: bit EPI = 64-row tiles for that GEMM kind (-1 = off)
    int dtl_ffn = 2;          // BUSCA_DTL_FFN: 2 = the layer-wise path runs out-proj + norm1 + feed-forward + norm2 as ONE kernel, 1 = the feed-forward block only,
                              // 0 = one kernel per GEMM (H and x1 through HBM)
    int dtl_attn = 1;         // BUSCA_DTL_ATTN: 1 = QKV projection + attention of a (track, head) in one kernel where it is built (0: QKV GEMM + attention kernel)
    int dt_split = -1;        // BUSCA_DT_SPLIT: token-split tail of the fused kernel (two workgroups per track): -1 = when the last round of a launch would fill at most
                              // half of the CUs, 0 = never, 1 = as many of the last tracks as fit one round (tests)
    int dt_prof = 0;          // BUSCA_DT_PROF: phase stamps of the fused kernel (debug): 1 = the one-workgroup flavour, 2 = the token-split flavour (first tile's workgroup)
    int dt_exact_f32 = 0;     // 1 = a context loaded with BUSCA_PREC_F16X3 runs its forwards in exact float32 (the f32 fragment packing kept beside the split one):
                              // how the host re-runs a step whose x3 forward reported a clipped operand ("dt_status" 2)
    int crop_band = 1;        // BUSCA_CROP_BAND: 1 = crops through the LDS-staged band kernel (crop_band_kernel), 0 = one thread per output pixel (A/B, tests)
    int last_dt_grid = 0, last_dt_ntrk = 0, last_dt_split = 0;     // read-only: workgroups / tracks per workgroup / token-split tracks of the last fused launch
    static int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
    void from_env() {
        dt_ntrk = env_int("BUSCA_DT_NTRK", 0); dt_tiled = getenv("BUSCA_DT_TILED") != nullptr ? 1 : 0;
        dtl_rt = env_int("BUSCA_DTL_RT", 0); dtl_rt_mask = env_int("BUSCA_DTL_RT_MASK", -1);
        dt_prof = env_int("BUSCA_DT_PROF", 0); dt_split = env_int("BUSCA_DT_SPLIT", -1);
        dtl_ffn = env_int("BUSCA_DTL_FFN", 2); dtl_attn = env_int("BUSCA_DTL_ATTN", 1); crop_band = env_int("BUSCA_CROP_BAND", 1);
    }
};

struct busca_ctx {
    int device = 0;
    std::string err;
    BuscaOptions opt;
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
    std::set<const void*> lds_configured;   // kernels whose dynamic-LDS limit was raised on THIS device
    void* ecc_ws = nullptr; size_t ecc_ws_bytes = 0;    // busca_ecc_align scratch: 5 float images + partials
};

// Raise a kernel's dynamic LDS limit once per context (the attribute is per device, so a process driving several
// GPUs through several contexts must set it for each).
static int ensure_lds(busca_ctx* c, const void* kern, size_t bytes);

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

static int ensure_lds(busca_ctx* c, const void* kern, size_t bytes) {
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
    *out = c;
    return BUSCA_OK;
}

#ifndef BUSCA_BUILD_FLAGS
#define BUSCA_BUILD_FLAGS "unknown"
#endif
extern "C" const char* busca_build_info(void) { return "libbusca_hip gfx950 flags: " BUSCA_BUILD_FLAGS; }

// ReID schedule knobs by name ("reid_gram", "reid_halo", ...): the same fields the BUSCA_REID_* environment variables set when weights are
// loaded; through busca_set_option they change between two forwards of a loaded extractor (A/B runs, the tests that compare schedules).
static int* reid_option_field(ReidState& R, const std::string& n) {
    struct { const char* name; int* p; } tab[] = {
        {"reid_gram", &R.gram_mode}, {"reid_halo_min", &R.halo_min_blocks}, {"reid_halo_half", &R.halo_half_blocks},
        {"reid_halo_wpx", &R.halo_wpx}, {"reid_halo_wpx_min", &R.halo_wpx_min}, {"reid_gram_min", &R.gram_min_pixels}, {"reid_direct_rows", &R.direct_rows},
        {"reid_fuse_ds_layers", &R.fuse_ds_layers}, {"reid_fuse_c1_layers", &R.fuse_c1_layers}, {"reid_kwave_blocks", &R.kwave_blocks},
        {"reid_kwave_halo", &R.kwave_halo_blocks}, {"reid_kwave_nw", &R.kwave_nw}, {"reid_kwave_pt", &R.kwave_pt}, 
        {"reid_pipe_min", &R.pipe_min_tiles}, {"reid_pipe_half", &R.pipe_half_blocks},
        {"reid_x3_merge_layers", &R.x3_merge_layers}, {"reid_x3_half", &R.x3_half_blocks}, {"reid_x3_gram_min", &R.x3_gram_min}, {"reid_x3_merge_in_min", &R.x3_merge_in_min}, {"reid_x3_fuse_c1_min", &R.x3_fuse_c1_min}, {"reid_x3_narrow3", &R.x3_narrow3}, {"reid_x3_row3", &R.x3_row3}, {"reid_x3_ptail", &R.x3_ptail_min}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}
static bool* reid_option_flag(ReidState& R, const std::string& n) {
    struct { const char* name; bool* p; } tab[] = {{"reid_halo", &R.halo}, {"reid_fuse_c1", &R.fuse_c1}, {"reid_fuse_c1_small", &R.fuse_c1_small},
                                                   {"reid_stats2", &R.two_launch_stats}, {"reid_pipe_all", &R.pipe_all}, {"reid_x3_gram", &R.x3_gram}, {"reid_x3_merge_in", &R.x3_merge_in}, {"reid_x3_fuse_c1", &R.x3_fuse_c1}, {"reid_x3_stem_halo", &R.x3_stem_halo}, {"reid_x3_stem_u8", &R.x3_stem_u8}, {"reid_x3_stem_pool", &R.x3_stem_pool}};
    for (auto& e : tab) if (n == e.name) return e.p;
    return nullptr;
}

extern "C" int busca_set_option(busca_ctx* c, const char* name, int32_t value) {
    if (!c || !name) return BUSCA_EINVAL;
    BuscaOptions& o = c->opt;
    const std::string n(name);
    if (n.rfind("reid_", 0) == 0) {
        if (!c->reid.loaded) return fail(c, BUSCA_ENOWEIGHTS, "busca_set_option('%s'): ReID schedule options belong to a loaded extractor (load weights first)", name);
        if (int* p = reid_option_field(c->reid, n)) { *p = value; return BUSCA_OK; }
        if (bool* p = reid_option_flag(c->reid, n)) { *p = value != 0; return BUSCA_OK; }
        return fail(c, BUSCA_EINVAL, "busca_set_option: unknown option '%s'", name);
    }
    if (n == "dt_ntrk") o.dt_ntrk = value;
    else if (n == "dt_split") o.dt_split = value;
    else if (n == "dt_prof") o.dt_prof = value;
    else if (n == "dt_tiled") o.dt_tiled = value;
    else if (n == "dtl_rt") o.dtl_rt = value;
    else if (n == "dtl_rt_mask") o.dtl_rt_mask = value;
    else if (n == "dtl_ffn") o.dtl_ffn = value;
    else if (n == "dtl_attn") o.dtl_attn = value;
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
    if (n.rfind("reid_", 0) == 0) {
        if (int* p = reid_option_field(c->reid, n)) { *value = *p; return BUSCA_OK; }
        if (bool* p = reid_option_flag(c->reid, n)) { *value = *p ? 1 : 0; return BUSCA_OK; }
        return fail(c, BUSCA_EINVAL, "busca_get_option: unknown option '%s'", name);
    }
    if (n == "dt_ntrk") *value = o.dt_ntrk;
    else if (n == "dt_split") *value = o.dt_split;
    else if (n == "last_dt_split") *value = o.last_dt_split;
    else if (n == "dt_status") *value = c->dt.xerr ? *c->dt.xerr : 0;       // 0 ok, 1 a split launch lost a partner, 2 an x3 forward clipped an operand; valid once the forward's stream is
                                                                            // synchronised; cleared by busca_set_option("dt_status", 0) (an uncleared status is also returned by the next forward)
    else if (n == "dt_tiled") *value = o.dt_tiled;
    else if (n == "dtl_rt") *value = o.dtl_rt;
    else if (n == "dtl_rt_mask") *value = o.dtl_rt_mask;
    else if (n == "dtl_ffn") *value = o.dtl_ffn;
    else if (n == "dtl_attn") *value = o.dtl_attn;
    else if (n == "crop_band") *value = o.crop_band;
    else if (n == "dt_exact_f32") *value = o.dt_exact_f32;
    else if (n == "last_dt_grid") *value = o.last_dt_grid;
    else if (n == "last_dt_ntrk") *value = o.last_dt_ntrk;
    else return fail(c, BUSCA_EINVAL, "busca_get_option: unknown option '%s'", name);
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
    if (c->dt.dev_tiled) hipFree(c->dt.dev_tiled);
    if (c->dt.ws) hipFree(c->dt.ws);
    if (c->dt.dev_blob32) hipFree(c->dt.dev_blob32);
    if (c->dt.xch) hipFree(c->dt.xch);
    if (c->dt.xflag) hipFree(c->dt.xflag);
    if (c->dt.xlg) hipFree(c->dt.xlg);
    if (c->dt.xerr) hipHostFree(c->dt.xerr);
    if (c->crop_fill) hipFree(c->crop_fill);
    if (c->ecc_ws) hipFree(c->ecc_ws);
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
    P.dec_bias = dec_bias;
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
    {   // exchange buffers of the token-split tail: one round's worth of tracks (two workgroups each), sized for this width
        { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0) S.num_cu = prop.multiProcessorCount; }
        if (S.xch) { HIP_TRY(c, hipFree(S.xch)); S.xch = nullptr; }
        if (S.xflag) { HIP_TRY(c, hipFree(S.xflag)); S.xflag = nullptr; }
        if (S.xlg) { HIP_TRY(c, hipFree(S.xlg)); S.xlg = nullptr; }
        S.xslots = S.num_cu;
        const size_t per_slot = (size_t)2 * DT_XMAX_MT * 4 * (2 * (d / 64)) * 1024;       // parity x tile x wave x (K, V tiles of a head) x 64 lanes x 16 bytes
        if (d % 64 == 0) {
            HIP_TRY(c, hipMalloc(&S.xch, per_slot * S.xslots));
            HIP_TRY(c, hipMalloc((void**)&S.xflag, (size_t)S.xslots * DT_XMAX_MT * DT_XFLAGS * sizeof(unsigned)));
            HIP_TRY(c, hipMemset(S.xflag, 0, (size_t)S.xslots * DT_XMAX_MT * DT_XFLAGS * sizeof(unsigned)));
            HIP_TRY(c, hipMalloc((void**)&S.xlg, (size_t)S.xslots * DT_XMAX_MT * 16 * sizeof(float)));
            if (!S.xerr) {
                HIP_TRY(c, hipHostMalloc((void**)&S.xerr, sizeof(int), hipHostMallocMapped)); *S.xerr = 0;
                HIP_TRY(c, hipHostGetDevicePointer((void**)&S.xerr_dev, S.xerr, 0));
            }
            S.xepoch = 0;
            HIP_TRY(c, hipDeviceSynchronize());
        } else S.xslots = 0;
    }
    S.loaded = true;
    return BUSCA_OK;
}

// ---------------------------------------------------------------------------------------------------------
// How many of the last tracks of a launch run token-split (one workgroup per 16-token tile, `parts` per track), and how many tracks share a workgroup
// (*pair: 1 or 2).  The tracks of the last, partial round of one-track workgroups are spread over the CUs that round would leave idle: a single 32-track
// step of three tiles runs on 96 CUs instead of 32 (0.46 -> 0.22 ms); 640 such tracks on 256 CUs are two rounds + 192 workgroups holding one tile index
// of two tracks each (0.35 ms) instead of three rounds.  One track per workgroup is the faster flavour while its workgroups fit ONE pass over the CUs
// (a one-tile workgroup is bound by its weight stream through the CU's vector memory path: two to a CU take twice as long); beyond that two tracks
// share a workgroup, every streamed weight fragment feeding two tiles (f32 flavour, tracks of three tiles or more - with two tiles such a workgroup
// would do a whole track's work).  A partial round too large for either stays one workgroup per track.
static int dt_split_tracks(const busca_ctx* c, int B, int parts, int prec, bool can_pair, int* pair) {
    const DTState& S = c->dt;
    *pair = 1;
    if (S.xslots <= 0 || c->opt.dt_split == 0 || parts > DT_XMAX_MT) return 0;
    if (c->opt.dt_split > 0) {                // tests: 1 = one track per workgroup, 2 = two
        *pair = (c->opt.dt_split == 2 && can_pair) ? 2 : 1;
        return std::min(B, S.xslots / 2);
    }
    // (f16 is left alone: that kernel is bound by the weight stream, which every workgroup of a split track repeats - measured 0.083 vs 0.085 ms for a
    // 32-track step, slower from one pass of workgroups on)
    if (prec == BUSCA_PREC_F16) return 0;
    const int rem = B % S.num_cu;
    if (rem == 0 || rem + 1 > S.xslots) return 0;
    // x3 is bound by the weight stream (L1 / L2), not by the MFMA: a partial round already runs faster than a full one (640 tracks 0.607 ms against
    // 3 x 0.225) and split workgroups add weight traffic - they pay only while the launch is small (32-track step 0.183 -> 0.134 ms; 128 tracks of two
    // tiles at d = 512: 0.407 -> 0.430)
    if (prec == BUSCA_PREC_F16X3) return 2 * parts * rem <= S.num_cu ? rem : 0;
    if (parts * rem <= S.num_cu) return rem;
    if (can_pair && parts * ((rem + 1) / 2) <= S.num_cu) { *pair = 2; return rem; }
    return 0;
}

static int dt_prof_report(busca_ctx* c, long long* d, int nwg, hipStream_t s) {
    HIP_TRY(c, hipStreamSynchronize(s));
    long long h[4 * DT_PROF_SLOTS];
    HIP_TRY(c, hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipFree(d));
    fprintf(stderr, "DT_PROF grid=%d:", nwg);
    for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "\n w%d", w);
        for (int i = 1; i < DT_PROF_SLOTS; ++i)
            if (h[w * DT_PROF_SLOTS + i]) fprintf(stderr, " %d:%lld", i, h[w * DT_PROF_SLOTS + i] - h[w * DT_PROF_SLOTS]);
    }
    fprintf(stderr, "\n");
    return BUSCA_OK;
}

template <int PREC, int MT, int D, int FF, int NCH, int NTRK>
static int dt_launch_split(busca_ctx* c, const DTParams& P0, int nsplit, hipStream_t s) {
    typedef DTLds<PREC, 1, D, FF, 512, NCH, NTRK> LD;
    DTState& S = c->dt;
    auto kern = dt_fused_kernel<PREC, MT, D, FF, 512, NCH, NTRK, true>;
    const int nwg = ((nsplit + NTRK - 1) / NTRK) * MT;
    DTParams P = P0;
    P.nsingle = P.B - nsplit;
    P.xepoch = ++S.xepoch; P.xch = (unsigned long long*)S.xch; P.xflag = S.xflag; P.xlg = S.xlg; P.xerr = S.xerr_dev;
    { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
    if (c->opt.dt_prof == 2) {
        HIP_TRY(c, hipMalloc((void**)&P.prof, 4 * DT_PROF_SLOTS * sizeof(long long)));
        HIP_TRY(c, hipMemset(P.prof, 0, 4 * DT_PROF_SLOTS * sizeof(long long)));
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
        return dt_prof_report(c, P.prof, nwg, s);
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
    return BUSCA_OK;
}

template <int PREC, int MT, int D, int FF, int NCH, int NTRK = 1>
static int dt_launch(busca_ctx* c, const DTParams& P, hipStream_t s) {
    typedef DTLds<PREC, MT, D, FF, 512, NCH, NTRK> LD;
    static_assert(LD::TOTAL <= 160 * 1024, "LDS plan exceeds the 160 KiB of a CU");
    auto kern = dt_fused_kernel<PREC, MT, D, FF, 512, NCH, NTRK>;
    const int nwg = (P.B + NTRK - 1) / NTRK;
    c->opt.last_dt_grid = nwg; c->opt.last_dt_ntrk = NTRK; c->opt.last_dt_split = 0;
    if constexpr (MT >= 2 && MT <= DT_XMAX_MT && NTRK == 1) {
        constexpr bool PAIR = PREC != 1 && MT >= 3;       // the two-tracks-per-workgroup split flavour: f32 / x3, three tiles or more (with two tiles it would do a whole track's work)
        // (every flavour is configured by the first forward of a shape, whichever it takes: a later launch of another track count must not pay for it)
        if (c->dt.xslots > 0) {
            { int rc = ensure_lds(c, (const void*)dt_fused_kernel<PREC, MT, D, FF, 512, NCH, 1, true>, DTLds<PREC, 1, D, FF, 512, NCH, 1>::TOTAL); if (rc) return rc; }
            if constexpr (PAIR) { int rc = ensure_lds(c, (const void*)dt_fused_kernel<PREC, MT, D, FF, 512, NCH, 2, true>, DTLds<PREC, 1, D, FF, 512, NCH, 2>::TOTAL); if (rc) return rc; }
        }
        { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
        // whole rounds of one-track workgroups, then the tail's tracks one token tile per workgroup: ONE timed region (the step batch), two launches on the stream
        int pair = 1;
        const int nsplit = c->opt.dt_prof == 1 ? 0 : dt_split_tracks(c, P.B, MT, PREC, PAIR, &pair);
        if (nsplit > 0) {
            c->opt.last_dt_grid = P.B - nsplit + ((nsplit + pair - 1) / pair) * MT; c->opt.last_dt_split = nsplit; c->opt.last_dt_ntrk = pair;
            TimedLaunch tl(c, s);
            if (P.B > nsplit) hipLaunchKernelGGL(kern, dim3(P.B - nsplit), dim3(256), LD::TOTAL, s, P);
            int rc = BUSCA_OK;
            if constexpr (PAIR) { if (pair == 2) rc = dt_launch_split<PREC, MT, D, FF, NCH, 2>(c, P, nsplit, s); else rc = dt_launch_split<PREC, MT, D, FF, NCH, 1>(c, P, nsplit, s); }
            else rc = dt_launch_split<PREC, MT, D, FF, NCH, 1>(c, P, nsplit, s);
            if (rc) return rc;
            HIP_TRY(c, hipGetLastError());
            return BUSCA_OK;
        }
    }
    { int rc = ensure_lds(c, (const void*)kern, LD::TOTAL); if (rc) return rc; }
    const bool prof = c->opt.dt_prof == 1;   // debug: phase timestamps of workgroup 0
    if (prof) {
        DTParams Q = P;
        long long* d = nullptr;
        HIP_TRY(c, hipMalloc((void**)&d, 4 * DT_PROF_SLOTS * sizeof(long long)));
        HIP_TRY(c, hipMemset(d, 0, 4 * DT_PROF_SLOTS * sizeof(long long)));
        Q.prof = d;
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, Q);
        return dt_prof_report(c, d, nwg, s);
    }
    {
        TimedLaunch tl(c, s);
        hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LD::TOTAL, s, P);
    }
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

// ---- tiled (layer-wise) path ---------------------------------------------------------------------------------------
template <int PREC, int D, int EPI, int RT>
static int dtl_gemm_rt(busca_ctx* c, hipStream_t s, const DTLArgs& a, int ncolblocks) {
    constexpr int BM = 32 * RT;
    const size_t lds = (size_t)(BM + D) * 128 + 2 * 4 * BM * sizeof(float);
    auto kern = dtl_gemm_kernel<PREC, D, EPI, RT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(kern, dim3((a.M + BM - 1) / BM, ncolblocks), dim3(512), lds, s, a);
    return BUSCA_OK;
}

// 64-row tiles (two workgroups per CU: one's epilogue traffic overlaps the other's MFMA phase) for the epilogue-heavy GEMMs
// when the tile fits twice into the LDS (d <= 512); BUSCA_DTL_RT=4 / 2 forces either geometry.
template <int PREC, int D, int EPI>
static int dtl_gemm(busca_ctx* c, hipStream_t s, const DTLArgs& a, int ncolblocks) {
    const int rt_env = c->opt.dtl_rt, rt_mask = c->opt.dtl_rt_mask;      // bit EPI of the mask = 64-row tiles for that GEMM kind
    // default: 64-row tiles only when 128-row tiles would fill less than half the chip (fewer than 128 workgroups) - measured
    // 128 lost x 32 proposals x d512 (79 row blocks): 64-row tiles for the single-column-block GEMMs 0.93 -> 0.86 ms (f16), f32
    // 3.38 -> 2.9 ms; from 158 row blocks on (two such steps in flight) 64-row tiles LOSE 5-10 %, and at >= 256 workgroups
    // the two geometries are within +-8 % per GEMM kind with no consistent winner (512 x 64 x d512: 4.25 ms vs 4.6 ms)
    const bool underfilled = (long)((a.M + 127) / 128) * ncolblocks < 128;
    const bool small = rt_mask >= 0 ? ((rt_mask >> EPI) & 1) != 0 : (rt_env == 2 || (rt_env == 0 && underfilled));
    if (small && (size_t)(64 + D) * 128 + 2 * 4 * 64 * 4 <= 80 * 1024) return dtl_gemm_rt<PREC, D, EPI, 2>(c, s, a, ncolblocks);
    return dtl_gemm_rt<PREC, D, EPI, 4>(c, s, a, ncolblocks);
}

template <int PREC, int HD, int MT>
static int dtl_attention(busca_ctx* c, hipStream_t s, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    constexpr int ES = Prec<PREC>::ES, TPK = Prec<PREC>::CHUNK * Prec<PREC>::nchunks(MT);
    const size_t lds = (size_t)16 * MT * (HD * ES + 16) + (size_t)HD * (TPK * ES + 16);
    if (lds > 160 * 1024) return fail(c, BUSCA_EINVAL, "tiled attention: %d tokens x head dim %d do not fit the LDS in this precision", T, HD);
    auto kern = dtl_attention_kernel<PREC, HD, MT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(kern, dim3(B, NH), dim3(256), lds, s, qkv, O, T, D, NH, att);
    return BUSCA_OK;
}

template <int PREC, int HD>
static int dtl_attention_mt(busca_ctx* c, hipStream_t s, int MT, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    switch (MT) {
        case 1: return dtl_attention<PREC, HD, 1>(c, s, qkv, O, B, T, D, NH, att);
        case 2: return dtl_attention<PREC, HD, 2>(c, s, qkv, O, B, T, D, NH, att);
        case 3: return dtl_attention<PREC, HD, 3>(c, s, qkv, O, B, T, D, NH, att);
        case 4: return dtl_attention<PREC, HD, 4>(c, s, qkv, O, B, T, D, NH, att);
        case 5: return dtl_attention<PREC, HD, 5>(c, s, qkv, O, B, T, D, NH, att);
        case 6: return dtl_attention<PREC, HD, 6>(c, s, qkv, O, B, T, D, NH, att);
        case 7: return dtl_attention<PREC, HD, 7>(c, s, qkv, O, B, T, D, NH, att);
        case 8: return dtl_attention<PREC, HD, 8>(c, s, qkv, O, B, T, D, NH, att);
        case 9: return dtl_attention<PREC, HD, 9>(c, s, qkv, O, B, T, D, NH, att);
    }
    return fail(c, BUSCA_EINVAL, "tiled attention supports at most 144 tokens per track (got %d)", T);
}

// head width HD = d / nhead in {16, 32, 64, 128} (nhead 4 at d = 64 / 256 / 512 are 16 / 64 / 128)
template <int PREC>
static int dtl_attention_hd(busca_ctx* c, hipStream_t s, int MT, const void* qkv, void* O, int B, int T, int D, int NH, float* att) {
    switch (D / NH) {
        case 16: return dtl_attention_mt<PREC, 16>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 32: return dtl_attention_mt<PREC, 32>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 64: return dtl_attention_mt<PREC, 64>(c, s, MT, qkv, O, B, T, D, NH, att);
        case 128: return dtl_attention_mt<PREC, 128>(c, s, MT, qkv, O, B, T, D, NH, att);
    }
    return fail(c, BUSCA_EINVAL, "tiled attention: head width %d not built (16, 32, 64, 128)", D / NH);
}

// QKV projection + attention of a (track, head) in one kernel (dtl_qkv_attn_kernel): built for the shipped head geometry (four heads: d = 512 /
// 128-wide, d = 256 / 64-wide) and the token counts the one-kernel path cannot hold.  Returns false when this shape is not built.
template <int PREC, int D, int HD, int MT>
static int dtl_qkv_attn_launch(busca_ctx* c, hipStream_t s, const DTLQkvAttnArgs& a, int B) {
    constexpr int ES = Prec<PREC>::ES, CH = Prec<PREC>::CHUNK, TP = 16 * MT, TPK = CH * Prec<PREC>::nchunks(MT);
    constexpr size_t ga = (size_t)TP * ((D / 2) * ES + 16), at = (size_t)2 * TP * (HD * ES + 16) + (size_t)HD * (TPK * ES + 16);
    constexpr size_t lds = ga > at ? ga : at;
    static_assert(lds <= 160 * 1024, "fused QKV + attention: LDS plan");
    auto kern = dtl_qkv_attn_kernel<PREC, D, HD, MT>;
    { int rc = ensure_lds(c, (const void*)kern, lds); if (rc) return rc; }
    TimedLaunch tl(c, s);
    hipLaunchKernelGGL(kern, dim3(B, a.NH), dim3(4 * HD), lds, s, a);
    return BUSCA_OK;
}
template <int PREC, int D>
static bool dtl_qkv_attn(busca_ctx* c, hipStream_t s, const DTLQkvAttnArgs& a, int B, int MT, int* rc) {
    constexpr int HD = D / 4;
    *rc = BUSCA_OK;
    if (D < 256 || a.NH != 4) return false;
#define QA(M_) case M_: *rc = dtl_qkv_attn_launch<PREC, (D >= 256 ? D : 256), (D >= 256 ? HD : 64), M_>(c, s, a, B); return true
    if constexpr (PREC == 1) { switch (MT) { QA(5); QA(6); QA(7); QA(8); QA(9); } }
    else { switch (MT) { QA(3); QA(4); QA(5); } }
#undef QA
    return false;
}

// Workspace of the layer-wise path for M rows (bytes).  Grown outside the forward by busca_dt_reserve; a forward that finds
// it too small grows it itself (one stream synchronisation + hipMalloc, first call of a larger shape only).
static size_t dtl_ws_bytes(size_t M, int D, int FF, size_t es) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    return al(M * D * 4) + al(M * D * 2) + al(M * 3 * D * es) + al(M * D * es) + al(M * FF * es) + al(M * 3 * 4);
}

static int dtl_ws_ensure(busca_ctx* c, size_t need, hipStream_t s) {
    DTState& S = c->dt;
    if (S.ws_bytes >= need) return BUSCA_OK;
    if (S.ws) { HIP_TRY(c, hipStreamSynchronize(s)); HIP_TRY(c, hipFree(S.ws)); S.ws = nullptr; S.ws_bytes = 0; }
    if (hipMalloc(&S.ws, need) != hipSuccess) return fail(c, BUSCA_ENOMEM, "cannot allocate %zu bytes of DT workspace", need);
    S.ws_bytes = need;
    return BUSCA_OK;
}

template <int PREC, int D>
static int dt_forward_tiled(busca_ctx* c, const DTParams& K, hipStream_t s) {
    DTState& S = c->dt;
    constexpr size_t ES = Prec<PREC>::ES;
    const int T = K.T, B = K.B, L = K.L, P = K.P, FF = S.cfg.ff, NH = S.cfg.nhead, E = 512;
    const int MT = (T + 15) / 16;
    if (MT > 9) return fail(c, BUSCA_EINVAL, "tiled DT path supports at most 144 tokens per track (T=%d)", T);
    if (P + K.nspec > 128) return fail(c, BUSCA_EINVAL, "tiled DT path supports at most %d proposals (P=%d)", 128 - K.nspec, P);
    const size_t M = (size_t)B * T;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    { int rc = dtl_ws_ensure(c, dtl_ws_bytes(M, D, FF, ES), s); if (rc) return rc; }
    char* p = (char*)S.ws;
    float* X = (float*)p; p += al(M * D * 4);
    _Float16* Xh = (_Float16*)p; p += al(M * D * 2);
    char* QKV = p; p += al(M * 3 * D * ES);
    char* O = p; p += al(M * D * ES);
    char* H = p; p += al(M * FF * ES);
    int* ids = (int*)p;
    const void* Xop = PREC == 0 ? (const void*)X : (const void*)Xh;      // GEMM operand copy of the residual stream
    { TimedLaunch tl(c, s); hipLaunchKernelGGL(dt_bucket_ids_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, K.mem_ltrb, K.can_ltrb, B, L, P, K.fake_f64, K.can_pos, K.nspec, K.sep_can, ids); }
    DTLArgs a{};
    a.M = (int)M; a.L = L; a.P = P; a.T = T; a.E = E; a.can_pos = K.can_pos; a.X = X; a.Xh = Xh; a.act = K.act;
    a.qscale = 1.0f / sqrtf((float)(D / NH));
    // embed + assembly + encoding
    a.W = S.tw.w_embed; a.K = E; a.bias = K.b_embed; a.mem_feat = K.mem_feat; a.can_feat = K.can_feat; a.ids = ids;
    a.lut_xy = K.lut_xy; a.lut_sz = K.lut_sz; a.lut_t = K.lut_t; a.lut_c = K.lut_c;
    a.tok_sep = K.tok_sep; a.tok_non = K.tok_non; a.tok_bad = K.tok_bad;
    { int rc = dtl_gemm<PREC, D, DTL_EPI_EMBED>(c, s, a, 1); if (rc) return rc; }
    for (int l = 0; l < K.nlayers; ++l) {
        const DTLayerW& W = K.layer[l];
        float* att = K.att ? K.att + (size_t)l * B * NH * T * T : nullptr;
        bool fused_attn = false;
        if (c->opt.dtl_attn != 0) {
            DTLQkvAttnArgs q{};
            q.Xop = Xop; q.w_in = W.w_in; q.b_in = W.b_in; q.O = O; q.att = att; q.T = T; q.NH = NH; q.qscale = a.qscale;
            int rc = BUSCA_OK;
            fused_attn = dtl_qkv_attn<PREC, D>(c, s, q, B, MT, &rc);
            if (rc) return rc;
        }
        if (!fused_attn) {
            a.A = Xop; a.lda = D; a.W = S.tw.w_in[l]; a.K = D; a.bias = W.b_in; a.out16 = QKV; a.ldo = 3 * D;
            {
                int rc = dtl_gemm<PREC, D, DTL_EPI_QKV>(c, s, a, 3);
                if (rc) return rc;
            }
            { int rc = dtl_attention_hd<PREC>(c, s, MT, QKV, O, B, T, D, NH, att); if (rc) return rc; }
        }
        const int ffn_mode = D >= 256 ? c->opt.dtl_ffn : 0;   // 2: out-proj + norm1 + feed-forward + norm2 in one kernel; 1: feed-forward block only; 0: layer-wise GEMMs
        if (ffn_mode != 2) {
            a.A = O; a.lda = D; a.W = S.tw.w_out[l]; a.K = D; a.bias = W.b_out; a.gamma = W.g1; a.beta = W.be1;
            { int rc = dtl_gemm<PREC, D, DTL_EPI_RESLN>(c, s, a, 1); if (rc) return rc; }
        }
        if (ffn_mode != 0) {
            // the row-local half of the layer in one kernel: x1 (mode 2) and H never reach HBM (dtl_ffn_kernel)
            DTLFfnArgs f{};
            f.Xop = Xop; f.Oop = O; f.X = X; f.Xh = Xh; f.w_out = W.w_out; f.w1 = W.w1; f.w2 = W.w2; f.b_out = W.b_out; f.g1 = W.g1; f.be1 = W.be1;
            f.b1 = W.b1; f.b2 = W.b2; f.gamma = W.g2; f.beta = W.be2; f.M = (int)M; f.FF = FF; f.act = K.act;
            constexpr int DK = D >= 256 ? D : 256;
            constexpr int BMF = DTLFfnGeom<PREC, DK>::BM;
            constexpr size_t flds = DTLFfnGeom<PREC, DK>::LDS;
            TimedLaunch tl(c, s);
            if (ffn_mode == 2) {
                auto kern = dtl_ffn_kernel<PREC, DK, true>;
                { int rc = ensure_lds(c, (const void*)kern, flds); if (rc) return rc; }
                hipLaunchKernelGGL(kern, dim3((unsigned)((M + BMF - 1) / BMF)), dim3(64 * DTLFfnGeom<PREC, DK>::NWV), flds, s, f);
            } else {
                auto kern = dtl_ffn_kernel<PREC, DK, false>;
                { int rc = ensure_lds(c, (const void*)kern, flds); if (rc) return rc; }
                hipLaunchKernelGGL(kern, dim3((unsigned)((M + BMF - 1) / BMF)), dim3(64 * DTLFfnGeom<PREC, DK>::NWV), flds, s, f);
            }
            continue;
        }
        a.A = Xop; a.lda = D; a.W = S.tw.w1[l]; a.K = D; a.bias = W.b1; a.out16 = H; a.ldo = FF;
        {
            int rc = dtl_gemm<PREC, D, DTL_EPI_FFN1>(c, s, a, FF / D);
            if (rc) return rc;
        }
        a.A = H; a.lda = FF; a.W = S.tw.w2[l]; a.K = FF; a.bias = W.b2; a.gamma = W.g2; a.beta = W.be2;
        { int rc = dtl_gemm<PREC, D, DTL_EPI_RESLN>(c, s, a, 1); if (rc) return rc; }
    }
    if (K.hidden) HIP_TRY(c, hipMemcpyAsync(K.hidden, X, M * D * sizeof(float), hipMemcpyDeviceToDevice, s));
    { TimedLaunch tl(c, s);
      hipLaunchKernelGGL((dtl_decoder_kernel<D>), dim3(B), dim3(256), 0, s, (const float*)X, T, L, P, K.can_pos, K.nspec, K.dec_g, K.dec_b, K.dec_w, K.dec_bias,
                         K.logits, K.probs, K.argmax); }
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
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
#define DT_CASE2(M, DD, NCH) if (fused_ok && two && prec == 1 && MT == M && d == DD) return dt_launch<1, M, DD, 2 * DD, NCH, 2>(c, K, s)
    DT_CASE2(3, 256, 1); DT_CASE2(2, 256, 1);        // d = 512: the parked f32 residual does not fit the LDS plan
#undef DT_CASE2
#define DT_CASE(PR, M, DD, NCH) if (fused_ok && prec == PR && MT == M && d == DD) return dt_launch<PR, M, DD, 2 * DD, NCH>(c, K, s)
    DT_CASE(0, 1, 64, 1); DT_CASE(0, 1, 256, 1); DT_CASE(0, 1, 512, 1); DT_CASE(1, 1, 64, 1); DT_CASE(1, 1, 256, 1); DT_CASE(1, 1, 512, 1);
    DT_CASE(0, 2, 64, 1); DT_CASE(0, 3, 64, 1); DT_CASE(0, 4, 64, 1);
    DT_CASE(0, 2, 256, 1); DT_CASE(0, 3, 256, 1);
    DT_CASE(0, 2, 512, 2);
    DT_CASE(1, 2, 64, 1); DT_CASE(1, 3, 64, 1); DT_CASE(1, 4, 64, 1);
    DT_CASE(1, 2, 256, 1); DT_CASE(1, 3, 256, 1); DT_CASE(1, 4, 256, 1); DT_CASE(1, 5, 256, 1);
    DT_CASE(1, 2, 512, 1); DT_CASE(1, 3, 512, 1); DT_CASE(1, 4, 512, 2);
    // split-fp16 (float32-equivalent) flavour: the f32 kernel's shapes
    DT_CASE(2, 1, 64, 1); DT_CASE(2, 1, 256, 1); DT_CASE(2, 1, 512, 1); DT_CASE(2, 2, 64, 1); DT_CASE(2, 3, 64, 1); DT_CASE(2, 4, 64, 1);
    DT_CASE(2, 2, 256, 1); DT_CASE(2, 3, 256, 1); DT_CASE(2, 2, 512, 2);
#undef DT_CASE
    // shapes beyond the fused kernel's on-chip plan: layer-wise tiled path, same arithmetic type
    if (prec == BUSCA_PREC_F16) {       // (x3 beyond the fused kernel's shapes: the exact f32 layer-wise path)
        if (d == 64) return dt_forward_tiled<1, 64>(c, K, s);
        if (d == 256) return dt_forward_tiled<1, 256>(c, K, s);
        if (d == 512) return dt_forward_tiled<1, 512>(c, K, s);
    } else {
        if (prec == BUSCA_PREC_F16X3) {      // the f32 packing of the matrices (proto32)
            K.w_embed = c->dt.proto32.w_embed;
            for (int l = 0; l < K.nlayers; ++l) { K.layer[l].w_in = c->dt.proto32.layer[l].w_in; K.layer[l].w_out = c->dt.proto32.layer[l].w_out; K.layer[l].w1 = c->dt.proto32.layer[l].w1; K.layer[l].w2 = c->dt.proto32.layer[l].w2; }
        }
        if (d == 64) return dt_forward_tiled<0, 64>(c, K, s);
        if (d == 256) return dt_forward_tiled<0, 256>(c, K, s);
        if (d == 512) return dt_forward_tiled<0, 512>(c, K, s);
    }
    return fail(c, BUSCA_EINVAL, "no Decision-Transformer kernel for T=%d (tiles %d), d=%d", K.T, MT, d);
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
    const int n = B * (L + 2 * (P + nspec));
    hipLaunchKernelGGL(dt_bucket_ids_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, mem_ltrb, can_ltrb, B, L, P, fake64, can_pos, nspec, sep_can, ids);
    HIP_TRY(c, hipGetLastError());
    return BUSCA_OK;
}

#include "capi_geometry.hip.inc"
#include "capi_reid.hip.inc"
