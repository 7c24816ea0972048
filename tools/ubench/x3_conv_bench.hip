// Stand-alone correctness + timing harness of conv_x3_kernel (busca_amd/csrc/reid_x3.hip.inc): every conv shape of a 512-crop ReID pass,
// each tile configuration timed on the same data, sampled outputs against a float64 host evaluation of the float32 operands.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/x3_conv_bench.hip -o tools/ubench/x3_conv_bench
// Run (GPU box):  tools/ubench/x3_conv_bench [case ...]
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define BUSCA_PREC_F16 1
#define BUSCA_PREC_F32 0
#include "../../busca_amd/csrc/reid_kernel.hip.inc"
#include "../../busca_amd/csrc/reid_x3.hip.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Case { const char* name; int n, H, W, Cin, Cout, k, stride, stg, epi; };
static const Case CASES[] = {
    {"L1conv1 1x1 256->64 plain   ", 512, 96, 32, 256, 64, 1, 1, X3_PLAIN, X3_RAW},
    {"L1conv2 3x3 64->64 bn       ", 512, 96, 32, 64, 64, 3, 1, X3_BN, X3_RAW},
    {"L1conv3 1x1 64->256 bn stats", 512, 96, 32, 64, 256, 1, 1, X3_BN, X3_STATS},
    {"L1conv3 1x1 64->256 bn merge", 512, 96, 32, 64, 256, 1, 1, X3_BN, X3_MERGE},
    {"L2conv1 1x1 512->128 plain  ", 512, 48, 16, 512, 128, 1, 1, X3_PLAIN, X3_RAW},
    {"L2conv2 3x3 128->128 bn     ", 512, 48, 16, 128, 128, 3, 1, X3_BN, X3_RAW},
    {"L2conv3 1x1 128->512 bn stat", 512, 48, 16, 128, 512, 1, 1, X3_BN, X3_STATS},
    {"L2conv3 1x1 128->512 bn merg", 512, 48, 16, 128, 512, 1, 1, X3_BN, X3_MERGE},
    {"L3conv1 1x1 1024->256 plain ", 512, 24, 8, 1024, 256, 1, 1, X3_PLAIN, X3_RAW},
    {"L3conv2 3x3 256->256 bn     ", 512, 24, 8, 256, 256, 3, 1, X3_BN, X3_RAW},
    {"L3conv3 1x1 256->1024 bn sta", 512, 24, 8, 256, 1024, 1, 1, X3_BN, X3_STATS},
    {"L3conv3 1x1 256->1024 bn mer", 512, 24, 8, 256, 1024, 1, 1, X3_BN, X3_MERGE},
    {"L4conv1 1x1 2048->512 plain ", 512, 12, 4, 2048, 512, 1, 1, X3_PLAIN, X3_RAW},
    {"L4conv2 3x3 512->512 bn     ", 512, 12, 4, 512, 512, 3, 1, X3_BN, X3_RAW},
    {"L4conv3 1x1 512->2048 bn mer", 512, 12, 4, 512, 2048, 1, 1, X3_BN, X3_MERGE},
    {"L2b0c2  3x3s2 128->128 bn   ", 512, 96, 32, 128, 128, 3, 2, X3_BN, X3_RAW},
    {"L2ds    1x1s2 256->512 plain", 512, 96, 32, 256, 512, 1, 2, X3_PLAIN, X3_RAW},
    {"L1conv3 1x1 64->256 bn raw  ", 512, 96, 32, 64, 256, 1, 1, X3_BN, X3_RAW},
    {"stem    7x7s2 3->64         ", 512, 384, 128, 3, 64, 7, 2, X3_STEM, X3_RAW},
    {"small   3x3 64->256 bn tail ", 3, 13, 7, 64, 256, 3, 1, X3_BN, X3_RAW},
    {"small   1x1s2 128->256 plain", 5, 9, 7, 128, 256, 1, 2, X3_PLAIN, X3_RAW},
    {"L3conv3 1x1 256->1024 bn raw", 512, 24, 8, 256, 1024, 1, 1, X3_BN, X3_RAW},
    {"L4conv3 1x1 512->2048 bn raw", 512, 12, 4, 512, 2048, 1, 1, X3_BN, X3_RAW},
    {"L4conv3 1x1 512->2048 bn sta", 512, 12, 4, 512, 2048, 1, 1, X3_BN, X3_STATS},
};

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static inline float frand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f; }

struct Variant { const char* name; int BM, BN; void (*launch)(const X3Args&, hipStream_t); bool (*ok)(const Case&); };
template <int WC, int WP, int CT, int PT, int STG, int KS, int EPI, int EXP = 0, int SKEW = 0>
static void launch_one(const X3Args& a, hipStream_t s) {
    static bool configured = false;
    if (!configured) { CK(hipFuncSetAttribute((const void*)conv_x3_kernel<WC, WP, CT, PT, STG, KS, EPI, EXP, SKEW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<WC, WP, CT, PT>(2048, STG == X3_BN))); configured = true; }
    const unsigned nb = (unsigned)(((a.gridM + 7) / 8) * 8 * a.gridN);
    const size_t lds = x3_lds_bytes<WC, WP, CT, PT>(a.Cin, STG == X3_BN);
    hipLaunchKernelGGL((conv_x3_kernel<WC, WP, CT, PT, STG, KS, EPI, EXP, SKEW>), dim3(nb), dim3(64 * WC * WP), lds, s, a);
}
// dispatch a tile configuration over the (STG, KS, EPI) combinations the ReID schedule uses
template <int WC, int WP, int CT, int PT, int EXP = 0, int SKEW = 0>
static void launch_tile(const X3Args& a, hipStream_t s, int stg, int ks, int epi) {
#define LT(S_, K_, E_) if (stg == S_ && ks == K_ && epi == E_) { launch_one<WC, WP, CT, PT, S_, K_, E_, EXP, SKEW>(a, s); return; }
    LT(X3_PLAIN, 1, X3_RAW) LT(X3_BN, 1, X3_RAW) LT(X3_BN, 3, X3_RAW) LT(X3_BN, 1, X3_STATS) LT(X3_BN, 1, X3_MERGE) LT(X3_STEM, 7, X3_RAW)
#undef LT
    fprintf(stderr, "no instantiation stg=%d ks=%d epi=%d\n", stg, ks, epi); exit(1);
}
static int g_stg, g_ks, g_epi;
#define VARIANT(WC_, WP_, CT_, PT_) {#WC_ "x" #WP_ " waves, " #CT_ "x" #PT_ " frags", 16 * PT_ * WP_, 16 * CT_ * WC_, [](const X3Args& a, hipStream_t s) { launch_tile<WC_, WP_, CT_, PT_>(a, s, g_stg, g_ks, g_epi); }, \
                                     [](const Case& c) { return c.Cout % (16 * CT_ * WC_) == 0 && (c.epi == X3_RAW || (16 * CT_ * WC_) >= 128); }}
#define VARIANT_EXP(WC_, WP_, CT_, PT_, X_) {#WC_ "x" #WP_ " waves " #CT_ "x" #PT_ " EXP " #X_, 16 * PT_ * WP_, 16 * CT_ * WC_, [](const X3Args& a, hipStream_t s) { launch_tile<WC_, WP_, CT_, PT_, X_>(a, s, g_stg, g_ks, g_epi); }, \
                                     [](const Case& c) { return c.Cout % (16 * CT_ * WC_) == 0 && (c.epi == X3_RAW || (16 * CT_ * WC_) >= 128); }}
#define VARIANT_SKEW(WC_, WP_, CT_, PT_, X_, S_) {#WC_ "x" #WP_ " waves " #CT_ "x" #PT_ " EXP " #X_ " SKEW " #S_, 16 * PT_ * WP_, 16 * CT_ * WC_, [](const X3Args& a, hipStream_t s) { launch_tile<WC_, WP_, CT_, PT_, X_, S_>(a, s, g_stg, g_ks, g_epi); }, \
                                     [](const Case& c) { return c.Cout % (16 * CT_ * WC_) == 0 && (c.epi == X3_RAW || (16 * CT_ * WC_) >= 128); }}
static const Variant VARIANTS[] = {
    VARIANT(8, 1, 2, 8),      // 128 px x 256 ch, 8 waves
    VARIANT(8, 1, 2, 4),      // 64 px x 256 ch, 8 waves (underfilled launches)
    VARIANT(4, 1, 2, 8),      // 128 px x 128 ch, 4 waves (two workgroups per CU)
    VARIANT(2, 2, 2, 4),      // 128 px x 64 ch, 4 waves
};

int main(int argc, char** argv) {
    const int ncases = sizeof(CASES) / sizeof(CASES[0]);
    for (int ci = 0; ci < ncases; ++ci) {
        if (argc > 1) { bool sel = false; for (int i = 1; i < argc; ++i) if (atoi(argv[i]) == ci) sel = true; if (!sel) continue; }
        const Case& c = CASES[ci];
        const bool stem = c.stg == X3_STEM;
        const int pad = c.k == 3 ? 1 : c.k == 7 ? 3 : 0;
        const int OH = (c.H + 2 * pad - c.k) / c.stride + 1, OW = (c.W + 2 * pad - c.k) / c.stride + 1;
        const int CinS = stem ? 4 : c.Cin;                 // stored channels
        const int M = c.n * OH * OW, taps = c.k * c.k, K = taps * c.Cin;
        const size_t nin = (size_t)c.n * c.H * c.W * CinS, nw = (size_t)c.Cout * K, nout = (size_t)M * c.Cout;
        std::vector<float> hin(nin), hw(nw), hss(2 * CinS), hss3(2 * c.Cout), hssd(2 * c.Cout), hidt;
        for (size_t i = 0; i < nin; ++i) hin[i] = (stem && (i & 3) == 3) ? 0.f : frand() * 2.0f;
        const float wsc = 1.0f / sqrtf((float)K);
        for (auto& v : hw) v = frand() * wsc * 1.7f;
        for (int co = 0; co < c.Cout; co += 7) for (int k = 0; k < K; ++k) hw[(size_t)co * K + k] *= 37.0f;       // uneven channel norms: the per-channel pre-scale matters
        for (int i = 0; i < CinS; ++i) { hss[2 * i] = 0.8f + 0.4f * frand(); hss[2 * i + 1] = 0.3f * frand(); }
        for (int i = 0; i < c.Cout; ++i) { hss3[2 * i] = 0.8f + 0.4f * frand(); hss3[2 * i + 1] = 0.3f * frand(); hssd[2 * i] = 0.9f + 0.3f * frand(); hssd[2 * i + 1] = 0.2f * frand(); }
        if (c.epi == X3_MERGE) { hidt.resize(nout); for (auto& v : hidt) v = frand() * 2.0f; }
        // hi / lo fragment order + per-channel descale, exactly as busca_reid_load_weights_ex packs them
        const int cch = stem ? 1 : c.Cin / 64, nhalf = stem ? 8 : 2 * taps * cch;
        std::vector<_Float16> hx3((size_t)c.Cout * nhalf * 64);
        std::vector<float> hinv(c.Cout);
        for (int co = 0; co < c.Cout; ++co) {
            float m = 0.f;
            for (int k = 0; k < K; ++k) m = std::max(m, std::fabs(hw[(size_t)co * K + k]));
            int ex = 0, kc = 0;
            if (m > 0.f) { std::frexp(m, &ex); kc = 13 - ex; }
            hinv[co] = std::ldexp(1.0f, -kc) / X3_XS;
            const int ct = co / 16, a = co % 16;
            for (int h = 0; h < nhalf; ++h)
                for (int b = 0; b < 4; ++b)
                    for (int e = 0; e < 8; ++e) {
                        float w;
                        if (stem) { const int kw = 2 * b + (e >> 2), chn = e & 3; w = (h < 7 && kw < 7 && chn < 3) ? hw[((size_t)co * taps + h * 7 + kw) * 3 + chn] : 0.f; }
                        else { const int st = h >> 1, kk = h & 1, tap = st / cch, chunk = st % cch; w = hw[((size_t)co * taps + tap) * c.Cin + chunk * 64 + kk * 32 + 8 * b + e]; }
                        const float ws = std::ldexp(w, kc);
                        const _Float16 hi = (_Float16)ws, lo = (_Float16)(ws - (float)hi);
                        const size_t base = (((size_t)ct * nhalf + h) * 2) * 512 + (size_t)(16 * b + a) * 8 + e;
                        hx3[base] = hi; hx3[base + 512] = lo;
                    }
        }
        float *din, *dout, *dout0, *dss, *dss3, *dssd, *dpart, *dpart0, *dzero, *dinv, *didt = nullptr; _Float16* dw;
        const int gridM = (M + 127) / 128, gridM64 = (M + 63) / 64;      // statistics buffers sized for the 64-pixel tile variants
        CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&dw, hx3.size() * 2)); CK(hipMalloc(&dout, nout * 4)); CK(hipMalloc(&dout0, nout * 4));
        CK(hipMalloc(&dss, 8 * CinS)); CK(hipMalloc(&dss3, 8 * c.Cout)); CK(hipMalloc(&dssd, 8 * c.Cout)); CK(hipMalloc(&dinv, 4 * c.Cout));
        CK(hipMalloc(&dpart, (size_t)gridM64 * 2 * c.Cout * 4)); CK(hipMalloc(&dpart0, (size_t)gridM64 * 2 * c.Cout * 4));
        CK(hipMalloc(&dzero, 256)); CK(hipMemset(dzero, 0, 256));
        CK(hipMemcpy(din, hin.data(), nin * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hx3.data(), hx3.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dss, hss.data(), 8 * CinS, hipMemcpyHostToDevice)); CK(hipMemcpy(dss3, hss3.data(), 8 * c.Cout, hipMemcpyHostToDevice));
        CK(hipMemcpy(dssd, hssd.data(), 8 * c.Cout, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, hinv.data(), 4 * c.Cout, hipMemcpyHostToDevice));
        if (c.epi == X3_MERGE) { CK(hipMalloc(&didt, nout * 4)); CK(hipMemcpy(didt, hidt.data(), nout * 4, hipMemcpyHostToDevice)); }
        const bool idt_bn = c.epi == X3_MERGE && (ci & 1);
        unsigned long long* dts; CK(hipMalloc(&dts, 2048 * 8 * 8 * 8)); CK(hipMemset(dts, 0, 2048 * 8 * 8 * 8));

        X3Args g{};
        g.in = din; g.in_ss = c.stg == X3_BN ? dss : nullptr; g.w = dw; g.inv = dinv; g.out = dout; g.partials = dpart; g.zero = dzero; g.wts = nullptr;
        g.out_ss = dss3; g.idt = didt; g.idt_ss = idt_bn ? dssd : nullptr; g.ts = dts;
        g.M = M; g.Cin = CinS; g.Cout = c.Cout; g.H = c.H; g.W = c.W; g.OH = OH; g.OW = OW; g.stride = c.stride; g.pad = pad; g.OHWo = OH * OW;
        g_stg = c.stg; g_ks = c.k; g_epi = c.epi;

        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_it = [&](auto&& fn, int iters) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                                                   float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / iters; };
        const double flops = 2.0 * M * c.Cout * (double)K;
        const double gb = ((double)nin * 4 + (c.epi == X3_STATS ? 0.0 : (double)nout * 4) + (c.epi == X3_MERGE ? (double)nout * 4 : 0.0)) / 1e9;
        std::vector<float> hout(nout), hout_v, hpart((size_t)gridM * 2 * c.Cout), hpart_v;
        bool first = true;
        printf("[%2d] %s M=%7d K=%5d  %.1f GFLOP, %.2f GB compulsory\n", ci, c.name, M, K, flops / 1e9, gb);
        for (const Variant& v : VARIANTS) {
            if (!v.ok(c)) continue;
            g.gridM = (M + v.BM - 1) / v.BM; g.gridN = c.Cout / v.BN;
            g.out = first ? dout0 : dout; g.partials = first ? dpart0 : dpart;
            if (c.epi != X3_STATS) CK(hipMemset(g.out, 0xff, nout * 4));
            const double us = time_it([&] { v.launch(g, 0); }, 10);
            CK(hipGetLastError()); CK(hipDeviceSynchronize());
            const char* verdict = "";
            if (first) {
                if (c.epi != X3_STATS) CK(hipMemcpy(hout.data(), dout0, nout * 4, hipMemcpyDeviceToHost));
                if (c.epi != X3_MERGE) CK(hipMemcpy(hpart.data(), dpart0, hpart.size() * 4, hipMemcpyDeviceToHost));
            } else {          // every tile configuration walks K in the same order: bit-identical outputs (statistics: same per 128-pixel tile when BM is 128)
                bool same = true;
                if (c.epi != X3_STATS) { hout_v.resize(nout); CK(hipMemcpy(hout_v.data(), dout, nout * 4, hipMemcpyDeviceToHost)); same = memcmp(hout_v.data(), hout.data(), nout * 4) == 0; }
                if (c.epi != X3_MERGE && v.BM == 128) { hpart_v.resize(hpart.size()); CK(hipMemcpy(hpart_v.data(), dpart, hpart.size() * 4, hipMemcpyDeviceToHost)); same = same && memcmp(hpart_v.data(), hpart.data(), hpart.size() * 4) == 0; }
                verdict = same ? " == first" : " DIFFERS FROM FIRST";
            }
            printf("      %-24s tiles %6d: %8.1f us  %6.1f TF f32-equivalent (%6.0f raw f16)  %5.2f TB/s%s\n", v.name, g.gridM * g.gridN, us, flops / us / 1e6, 3 * flops / us / 1e6, gb / us * 1e3, verdict);
            first = false;
            if (strstr(v.name, "EXP 9")) {
                std::vector<unsigned long long> hts(2048 * 8 * 8);
                CK(hipMemcpy(hts.data(), dts, hts.size() * 8, hipMemcpyDeviceToHost));
                const int nwg = std::min(2048, ((g.gridM + 7) / 8) * 8 * g.gridN), nst = c.stg == X3_STEM ? 4 : taps * cch, nstr = (nst + 1) / 2 * 2;
                for (int half = 0; half < 2; ++half) {
                    double d[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int cnt = 0;
                    for (int wg = 0; wg < nwg; ++wg) for (int w = 4 * half; w < 4 * half + 4; ++w) { const unsigned long long* r = &hts[((size_t)wg * 8 + w) * 8]; if (!r[6]) continue; ++cnt; for (int k = 0; k < 8; ++k) d[k] += (double)r[k]; }
                    if (cnt) printf("        waves %d-%d, cycles per K step: wait A %.0f | transform + ds_write %.0f | issue A loads %.0f | first fragments %.0f | MFMAs + reads %.0f | issue W loads %.0f | barrier %.0f   (sum %.0f)\n",
                                    4 * half, 4 * half + 3, d[0] / cnt / nstr, d[1] / cnt / nstr, d[2] / cnt / nstr, d[3] / cnt / nstr, d[4] / cnt / nstr, d[5] / cnt / nstr, d[6] / cnt / nstr,
                                    (d[0] + d[1] + d[2] + d[3] + d[4] + d[5] + d[6]) / cnt / nstr);
                }
            }
        }
        // ---- check of the first variant: sampled outputs against float64
        auto opnd = [&](size_t idx, int ch) -> double {
            const float x = hin[idx];
            if (c.stg != X3_BN) return (double)x;
            return (double)fmaxf(fmaf(x, hss[2 * ch], hss[2 * ch + 1]), 0.f);
        };
        double maxerr = 0, maxrel = 0; int bad = 0;
        const int nsamp = 2000;
        std::vector<double> conv_of((size_t)0);
        for (int sidx = 0; sidx < nsamp && c.epi != X3_STATS; ++sidx) {
            rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
            int m = (int)(rs % (unsigned long long)M); int co = (int)((rs >> 32) % (unsigned long long)c.Cout);
            if (sidx < 64) m = M - 1 - sidx % (M < 64 ? M : 64);
            if (sidx >= 64 && sidx < 128) m = sidx - 64 < M ? sidx - 64 : 0;
            const int img = m / (OH * OW), rem = m % (OH * OW), oh = rem / OW, ow = rem % OW;
            double accd = 0, mag = 0;
            for (int kh = 0; kh < c.k; ++kh)
                for (int kw = 0; kw < c.k; ++kw) {
                    const int ih = oh * c.stride - pad + kh, iw = ow * c.stride - pad + kw;
                    if (ih < 0 || ih >= c.H || iw < 0 || iw >= c.W) continue;
                    const size_t base = (((size_t)img * c.H + ih) * c.W + iw) * CinS;
                    for (int ch = 0; ch < c.Cin; ++ch) { const double p = opnd(base + ch, ch) * (double)hw[((size_t)co * taps + kh * c.k + kw) * c.Cin + ch]; accd += p; mag += fabs(p); }
                }
            double want = accd;
            if (c.epi == X3_MERGE) {
                double id = hidt[(size_t)m * c.Cout + co];
                if (idt_bn) id = (double)fmaf((float)id, hssd[2 * co], hssd[2 * co + 1]);
                want = fmax((double)fmaf((float)accd, hss3[2 * co], hss3[2 * co + 1]) + id, 0.0);
            }
            const double got = (double)hout[(size_t)m * c.Cout + co];
            const double err = fabs(got - want);
            maxerr = fmax(maxerr, err); maxrel = fmax(maxrel, err / (mag + 1e-30));
            if (err > 1e-6 * mag + 1e-6) { if (++bad < 5) fprintf(stderr, "  mismatch m=%d co=%d got %.9g want %.9g (sum|ab| %.3g)\n", m, co, got, want, mag); }
        }
        double smax = 0;
        for (int t = 0; t < 3 && c.epi == X3_RAW; ++t) {
            const int co = (t * 37) % c.Cout, mt = t == 0 ? gridM - 1 : (t * 11) % gridM;
            double s1 = 0, s2 = 0;
            for (int r = 0; r < 128; ++r) { const int m = mt * 128 + r; if (m >= M) break; const double v = (double)hout[(size_t)m * c.Cout + co]; s1 += v; s2 += v * v; }
            smax = fmax(smax, fabs(s1 - hpart[((size_t)mt * 2 + 0) * c.Cout + co]) / (1.0 + fabs(s1)));
            smax = fmax(smax, fabs(s2 - hpart[((size_t)mt * 2 + 1) * c.Cout + co]) / (1.0 + fabs(s2)));
        }
        printf("      check: max |err| %.2e, max err / sum|ab| %.2e, bad %d, statistics rel %.1e  %s\n", maxerr, maxrel, bad, smax, (bad == 0 && smax < 1e-4) ? "OK" : "FAIL");
        fflush(stdout);
        hipFree(din); hipFree(dw); hipFree(dout); hipFree(dout0); hipFree(dss); hipFree(dss3); hipFree(dssd); hipFree(dpart); hipFree(dpart0); hipFree(dzero); hipFree(dinv);
        if (didt) hipFree(didt);
    }
    return 0;
}
