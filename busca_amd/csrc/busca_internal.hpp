// Internal header of libbusca_hip.so: what its translation units share (the context, the options, error / timing helpers, the entry points one unit
// offers the others).  The library is built from several units compiled in parallel (busca_amd/build.py):
//   busca_hip.hip      context, options, timing, Decision-Transformer weight packing + dispatch, geometry / crop / tracking kernels and their C-ABI
//   busca_dt_f32 / _f16 / _x3.hip   the fused Decision-Transformer kernel, one unit per arithmetic flavour   (+ busca_dt_aux.hip, see there)
//   busca_dtl_f32 / _f16 / _x3.hip  the layer-wise Decision-Transformer path
//   busca_reid.hip     the ReID extractor (every flavour) and its C-ABI
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <type_traits>
#include <vector>

#pragma GCC visibility push(default)      // the library is built with -fvisibility=hidden: its exports are exactly what include/busca_hip.h declares
#include "../../include/busca_hip.h"
#pragma GCC visibility pop
#include "dt_types.hpp"

struct ReidState;                   // reid_kernel.hip.inc (complete only inside busca_reid.hip)

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
    ReidState* reid = nullptr;     // owned by the ReID translation unit (busca_reid.hip: reid_state_new / reid_state_delete)
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
int ensure_lds(busca_ctx* c, const void* kern, size_t bytes);

static inline int fail(busca_ctx* c, int code, const char* fmt, ...) {
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


// ---- kernel timing (busca_hip.hip) ----------------------------------------------------------------------------------
void timing_drain(busca_ctx* c);
hipEvent_t timing_event(busca_ctx* c);
struct TimedLaunch {   // RAII: records start/stop events around one kernel launch when timing is on
    busca_ctx* c; hipStream_t s; hipEvent_t e0{}, e1{}; bool on;
    TimedLaunch(busca_ctx* c_, hipStream_t s_) : c(c_), s(s_), on(c_->timing) {
        if (on) { e0 = timing_event(c); e1 = timing_event(c); hipEventRecord(e0, s); }
    }
    ~TimedLaunch() {
        if (on) { hipEventRecord(e1, s); c->ev_pending.emplace_back(e0, e1); if (c->ev_pending.size() > 4096) timing_drain(c); }
    }
};

// ---- what the units offer each other ----------------------------------------------------------------------------------
// fused Decision-Transformer kernel (busca_dt_*.hip): launches the instantiation built for (MT token tiles, width d) in that flavour, or returns BUSCA_ENOKERNEL
// when none is built (the caller then takes the layer-wise path).  `two`: the two-tracks-per-workgroup f16 flavour.
#define BUSCA_ENOKERNEL 1000         // internal, never returned through the C-ABI
int dt_fused_f32(busca_ctx* c, const DTParams& K, int MT, int d, hipStream_t s);
int dt_fused_f16(busca_ctx* c, const DTParams& K, int MT, int d, bool two, hipStream_t s);
int dt_fused_x3(busca_ctx* c, const DTParams& K, int MT, int d, hipStream_t s);
// layer-wise path (busca_dtl_*.hip)
int dt_tiled_f32(busca_ctx* c, const DTParams& K, int d, hipStream_t s);
int dt_tiled_f16(busca_ctx* c, const DTParams& K, int d, hipStream_t s);
int dt_tiled_x3(busca_ctx* c, const DTParams& K, int d, hipStream_t s);      // BUSCA_ENOKERNEL: no split-fp16 layer kernels for this width (the caller runs the exact f32 path)
int dt_split_ensure(busca_ctx* c);                 // allocates the token-split exchange buffers on first use (busca_hip.hip)
size_t dtl_ws_bytes(size_t M, int D, int FF, size_t es);
int dtl_ws_ensure(busca_ctx* c, size_t need, hipStream_t s);
void dt_bucket_ids_launch(busca_ctx* c, hipStream_t s, const float* mem_ltrb, const float* can_ltrb, int B, int L, int P, int fake_f64, int can_pos, int nspec, int sep_can, int* ids);
// ReID (busca_reid.hip)
ReidState* reid_state_new();
void reid_state_delete(ReidState* r);
bool reid_state_loaded(const ReidState* r);
int reid_set_option(busca_ctx* c, const char* name, int32_t value);          // "reid_*" names of busca_set_option / busca_get_option
int reid_get_option(busca_ctx* c, const char* name, int32_t* value);
