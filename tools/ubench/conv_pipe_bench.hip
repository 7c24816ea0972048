// Stand-alone correctness + timing harness of conv_pipe_kernel (busca_amd/csrc/reid_pipe.hip.inc) against the kernels it replaces.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/conv_pipe_bench.hip -o tools/ubench/conv_pipe_bench
// Run (GPU box):  tools/ubench/conv_pipe_bench [case ...]
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
#include "../../busca_amd/csrc/reid_wdirect.hip.inc"
#include "../../busca_amd/csrc/reid_pipe.hip.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Case { const char* name; int n, H, W, Cin, Cout, k, stride, stg; };
static const Case CASES[] = {
    {"L3conv1  1x1 1024->256  plain", 512, 24, 8, 1024, 256, 1, 1, STG_PLAIN},
    {"L3conv3  1x1 256->1024  bn   ", 512, 24, 8, 256, 1024, 1, 1, STG_BN},
    {"L3b0c2   3x3s2 256->256 bn   ", 512, 48, 16, 256, 256, 3, 2, STG_BN},
    {"L4conv2  3x3 512->512   bn   ", 512, 12, 4, 512, 512, 3, 1, STG_BN},
    {"L4conv3  1x1 512->2048  bn   ", 512, 12, 4, 512, 2048, 1, 1, STG_BN},
    {"L4ds     1x1s2 1024->2048 pl ", 512, 24, 8, 1024, 2048, 1, 2, STG_PLAIN},
    {"L2b0c2   3x3s2 128->128 bn   ", 512, 96, 32, 128, 128, 3, 2, STG_BN},
    {"L3conv1m 1x1 1024->256  merge", 512, 24, 8, 1024, 256, 1, 1, STG_MERGE},
    {"L4conv1m 1x1 2048->512  merge", 512, 12, 4, 2048, 512, 1, 1, STG_MERGE},
    {"L2conv1m 1x1 512->128   merge", 512, 48, 16, 512, 128, 1, 1, STG_MERGE},
    {"L4b0c2   3x3s2 512->512 bn   ", 512, 24, 8, 512, 512, 3, 2, STG_BN},
    {"L3ds     1x1s2 512->1024 pl  ", 512, 48, 16, 512, 1024, 1, 2, STG_PLAIN},
    {"small    3x3 64->256 bn tail ", 3, 13, 7, 64, 256, 3, 1, STG_BN},
    {"L4conv2p 3x3 512->512   plain", 512, 12, 4, 512, 512, 3, 1, STG_PLAIN},
    {"L4b0c2p  3x3s2 512->512 plain", 512, 24, 8, 512, 512, 3, 2, STG_PLAIN},
    {"L3b0c2p  3x3s2 256->256 plain", 512, 48, 16, 256, 256, 3, 2, STG_PLAIN},
};

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static float frand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f; }

template <int NCT, int STG, int KS>
static void launch_pipe(const PipeArgs& g) {
    const unsigned nb = (unsigned)(((g.gridM + 7) / 8) * 8 * g.gridN);
    hipLaunchKernelGGL((conv_pipe_kernel<NCT, STG, KS>), dim3(nb), dim3(256), 0, 0, g);
}
template <int NCT, int STG, int KS>
static void launch_pipe_nosgb(const PipeArgs& g) {
    const unsigned nb = (unsigned)(((g.gridM + 7) / 8) * 8 * g.gridN);
    hipLaunchKernelGGL((conv_pipe_kernel<NCT, STG, KS, false>), dim3(nb), dim3(256), 0, 0, g);
}
static bool launch_pipe_nosgb_any(const PipeArgs& g, int nct, int stg, int ks) {
#define LP(N_, S_, K_) if (nct == N_ && stg == S_ && ks == K_) { launch_pipe_nosgb<N_, S_, K_>(g); return true; }
    LP(4, STG_BN, 1) LP(4, STG_BN, 3) LP(4, STG_MERGE, 1)
#undef LP
    return false;
}
static void launch_pipe_any(const PipeArgs& g, int nct, int stg, int ks) {
#define LP(N_, S_, K_) if (nct == N_ && stg == S_ && ks == K_) { launch_pipe<N_, S_, K_>(g); return; }
    LP(4, STG_PLAIN, 1) LP(4, STG_PLAIN, 3) LP(4, STG_BN, 1) LP(4, STG_BN, 3) LP(4, STG_MERGE, 1) LP(2, STG_BN, 3) LP(2, STG_MERGE, 1) LP(2, STG_PLAIN, 1) LP(2, STG_BN, 1)
#undef LP
    fprintf(stderr, "no instantiation nct=%d stg=%d ks=%d\n", nct, stg, ks); exit(1);
}

int main(int argc, char** argv) {
    const int ncases = sizeof(CASES) / sizeof(CASES[0]);
    for (int ci = 0; ci < ncases; ++ci) {
        if (argc > 1) { bool sel = false; for (int i = 1; i < argc; ++i) if (atoi(argv[i]) == ci) sel = true; if (!sel) continue; }
        const Case& c = CASES[ci];
        const int pad = c.k == 3 ? 1 : 0;
        const int OH = (c.H + 2 * pad - c.k) / c.stride + 1, OW = (c.W + 2 * pad - c.k) / c.stride + 1;
        const int M = c.n * OH * OW, K = c.k * c.k * c.Cin;
        const size_t nin = (size_t)c.n * c.H * c.W * c.Cin, nw = (size_t)c.Cout * K, nout = (size_t)M * c.Cout;
        std::vector<_Float16> hin(nin), hin2, hw(nw), hwkw(nw);
        std::vector<float> hss(2 * c.Cin), hss2(2 * c.Cin);
        for (auto& v : hin) v = (_Float16)(frand() * 2.0f);
        const float wsc = 1.0f / sqrtf((float)K);
        for (auto& v : hw) v = (_Float16)(frand() * wsc * 1.7f);
        for (int i = 0; i < c.Cin; ++i) { hss[2 * i] = 0.8f + 0.4f * frand(); hss[2 * i + 1] = 0.3f * frand(); hss2[2 * i] = 0.9f + 0.3f * frand(); hss2[2 * i + 1] = 0.2f * frand(); }
        if (c.stg == STG_MERGE) { hin2.resize(nin); for (auto& v : hin2) v = (_Float16)(frand() * 2.0f); }
        // fragment order [Cout/16][half step h = 2 (tap*chunks + chunk) + kk][lane = 16b + a][8] (capi_reid.hip.inc)
        {
            const int taps = c.k * c.k, cch = c.Cin / 64, nhalf = 2 * taps * cch;
            for (int ct = 0; ct < c.Cout / 16; ++ct)
                for (int h = 0; h < nhalf; ++h)
                    for (int ln = 0; ln < 64; ++ln)
                        for (int e = 0; e < 8; ++e) {
                            const int a = ln & 15, b = ln >> 4, st = h >> 1, kk = h & 1, tap = st / cch, chunk = st % cch;
                            const int co = 16 * ct + a, cin = chunk * 64 + kk * 32 + 8 * b + e;
                            hwkw[(((size_t)ct * nhalf + h) * 64 + ln) * 8 + e] = hw[((size_t)co * taps + tap) * c.Cin + cin];
                        }
        }
        _Float16 *din, *din2 = nullptr, *dw, *dwkw, *dout, *dout_old, *dmout = nullptr, *dzero;
        float *dss, *dss2, *dpart, *dpart_old;
        const int gridM = (M + 127) / 128;
        CK(hipMalloc(&din, nin * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dwkw, nw * 2)); CK(hipMalloc(&dout, nout * 2)); CK(hipMalloc(&dout_old, nout * 2));
        CK(hipMalloc(&dss, 8 * c.Cin)); CK(hipMalloc(&dss2, 8 * c.Cin)); CK(hipMalloc(&dpart, (size_t)gridM * 2 * c.Cout * 4)); CK(hipMalloc(&dpart_old, (size_t)gridM * 2 * c.Cout * 4));
        CK(hipMalloc(&dzero, 256)); CK(hipMemset(dzero, 0, 256));
        CK(hipMemcpy(din, hin.data(), nin * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwkw, hwkw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dss, hss.data(), 8 * c.Cin, hipMemcpyHostToDevice)); CK(hipMemcpy(dss2, hss2.data(), 8 * c.Cin, hipMemcpyHostToDevice));
        if (c.stg == STG_MERGE) { CK(hipMalloc(&din2, nin * 2)); CK(hipMemcpy(din2, hin2.data(), nin * 2, hipMemcpyHostToDevice)); CK(hipMalloc(&dmout, nin * 2)); CK(hipMemset(dmout, 0xff, nin * 2)); }
        CK(hipMemset(dout, 0xff, nout * 2));

        const int nct = c.Cout % 256 == 0 ? 4 : 2;
        PipeArgs g{};
        g.in = din; g.in_ss = c.stg == STG_PLAIN ? nullptr : dss; g.in2 = din2; g.in2_ss = (c.stg == STG_MERGE && (ci & 1)) ? dss2 : nullptr; g.mout = dmout;
        g.wkw = dwkw; g.out = dout; g.partials = dpart; g.zero = dzero; g.wts = nullptr;
        g.M = M; g.Cin = c.Cin; g.Cout = c.Cout; g.H = c.H; g.W = c.W; g.OH = OH; g.OW = OW; g.stride = c.stride; g.pad = pad; g.OHWo = OH * OW;
        g.gridM = gridM; g.gridN = c.Cout / (64 * nct);
        const bool tr2 = g.in2_ss != nullptr;

        ConvArgs o{};
        o.in = din; o.w = dw; o.in_ss = c.stg == STG_BN ? dss : nullptr; o.out = dout_old; o.partials = dpart_old; o.n = c.n; o.H = c.H; o.W = c.W; o.Cin = c.Cin;
        o.OH = OH; o.OW = OW; o.Cout = c.Cout; o.KH = c.k; o.KW = c.k; o.stride = c.stride; o.pad = pad; o.M = M; o.gridM = gridM; o.gridN = c.Cout == 64 ? 1 : c.Cout / 128;
        o.zero = dzero; o.OHWo = OH * OW; o.wkw = dwkw;
        const unsigned onb = (unsigned)(((gridM + 7) / 8) * 8 * o.gridN);

        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_it = [&](auto&& fn, int iters) { fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                                                   float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / iters; };
        const double flops = 2.0 * M * c.Cout * (double)K;
        const double us_new = time_it([&] { launch_pipe_any(g, nct, c.stg, c.k); }, 20);
        CK(hipGetLastError());
        double us_old = 0, us_wd = 0, us_nosgb = 0;
        if (launch_pipe_nosgb_any(g, nct, c.stg, c.k)) us_nosgb = time_it([&] { launch_pipe_nosgb_any(g, nct, c.stg, c.k); }, 20);
        { const double again = time_it([&] { launch_pipe_any(g, nct, c.stg, c.k); }, 20); printf("      pipe with interleave %.1f / %.1f us, without %.1f us\n", us_new, again, us_nosgb); }
        if (c.stg != STG_MERGE && c.Cout >= 128) us_old = time_it([&] { hipLaunchKernelGGL((conv_gemm64_kernel<2, CONV_NORMAL>), dim3(onb), dim3(256), 0, 0, o); }, 20);
        if (c.stg != STG_MERGE && c.k == 1 && c.stride == 1 && c.Cout % 256 == 0) {
            ConvArgs w = o; w.gridN = c.Cout / 256;
            const unsigned wb = (unsigned)(((gridM + 7) / 8) * 8 * w.gridN);
            us_wd = time_it([&] { hipLaunchKernelGGL((conv1x1_wd_kernel<CONV_NORMAL>), dim3(wb), dim3(256), 0, 0, w); }, 20);
        }
        CK(hipDeviceSynchronize());

        // ---- check: sampled outputs against a host evaluation in double of the same fp16 operands
        std::vector<_Float16> hout(nout), hmout;
        std::vector<float> hpart((size_t)gridM * 2 * c.Cout);
        CK(hipMemcpy(hout.data(), dout, nout * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hpart.data(), dpart, hpart.size() * 4, hipMemcpyDeviceToHost));
        if (c.stg == STG_MERGE) { hmout.resize(nin); CK(hipMemcpy(hmout.data(), dmout, nin * 2, hipMemcpyDeviceToHost)); }
        auto opnd = [&](size_t idx, int ch) -> _Float16 {      // staged operand value of input element idx (channel ch)
            const _Float16 x = hin[idx];
            if (c.stg == STG_PLAIN) return x;
            if (c.stg == STG_BN) return (_Float16)fmaxf(fmaf((float)x, hss[2 * ch], hss[2 * ch + 1]), 0.f);
            float id = (float)hin2[idx];
            if (tr2) id = fmaf(id, hss2[2 * ch], hss2[2 * ch + 1]);
            return (_Float16)fmaxf(fmaf((float)x, hss[2 * ch], hss[2 * ch + 1]) + id, 0.f);
        };
        double maxerr = 0, maxref = 0; int bad = 0;
        const int nsamp = 3000;
        for (int sidx = 0; sidx < nsamp; ++sidx) {
            rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17;
            int m = (int)(rs % (unsigned long long)M); int co = (int)((rs >> 32) % (unsigned long long)c.Cout);
            if (sidx < 64) m = M - 1 - sidx % (M < 64 ? M : 64);           // the tail rows of the last tile
            if (sidx >= 64 && sidx < 128) m = sidx - 64 < M ? sidx - 64 : 0;
            const int img = m / (OH * OW), rem = m % (OH * OW), oh = rem / OW, ow = rem % OW;
            double accd = 0;
            for (int kh = 0; kh < c.k; ++kh)
                for (int kw = 0; kw < c.k; ++kw) {
                    const int ih = oh * c.stride - pad + kh, iw = ow * c.stride - pad + kw;
                    if (ih < 0 || ih >= c.H || iw < 0 || iw >= c.W) continue;
                    const size_t base = (((size_t)img * c.H + ih) * c.W + iw) * c.Cin;
                    for (int ch = 0; ch < c.Cin; ++ch) accd += (double)(float)opnd(base + ch, ch) * (double)(float)hw[((size_t)co * c.k * c.k + kh * c.k + kw) * c.Cin + ch];
                }
            const double got = (double)(float)hout[(size_t)m * c.Cout + co];
            const double err = fabs(got - accd);
            maxerr = fmax(maxerr, err); maxref = fmax(maxref, fabs(accd));
            if (err > 2e-3 * fabs(accd) + 4e-3) { if (++bad < 5) fprintf(stderr, "  mismatch m=%d co=%d got %f want %f\n", m, co, got, accd); }
        }
        // merged tensor: every element exactly
        size_t mbad = 0;
        if (c.stg == STG_MERGE) {
            for (size_t i = 0; i < nin; ++i) {
                const _Float16 want = opnd(i, (int)(i % c.Cin));
                if (__builtin_bit_cast(unsigned short, want) != __builtin_bit_cast(unsigned short, hmout[i]) && !((float)want == 0.f && (float)hmout[i] == 0.f)) ++mbad;
            }
        }
        // statistics of a few channels: tile sums against the stored (fp16-rounded inputs, f32 accumulators) -> compare with sums of hout
        double smax = 0;
        for (int t = 0; t < 3; ++t) {
            const int co = (t * 37) % c.Cout, mt = t == 0 ? gridM - 1 : (t * 11) % gridM;
            double s1 = 0, s2 = 0;
            for (int r = 0; r < 128; ++r) { const int m = mt * 128 + r; if (m >= M) break; const double v = (double)(float)hout[(size_t)m * c.Cout + co]; s1 += v; s2 += v * v; }
            smax = fmax(smax, fabs(s1 - hpart[((size_t)mt * 2 + 0) * c.Cout + co]) / (1.0 + fabs(s1)));
            smax = fmax(smax, fabs(s2 - hpart[((size_t)mt * 2 + 1) * c.Cout + co]) / (1.0 + fabs(s2)));
        }
        printf("[%2d] %s M=%7d tiles=%5d: new %7.1f us %7.1f TF | gemm64 %7.1f us %6.1f TF | wd %7.1f us %6.1f TF | maxerr %.2e (ref %.1f) bad %d mergebad %zu statrel %.1e %s\n",
               ci, c.name, M, gridM * g.gridN, us_new, flops / us_new / 1e6, us_old, us_old > 0 ? flops / us_old / 1e6 : 0.0, us_wd, us_wd > 0 ? flops / us_wd / 1e6 : 0.0,
               maxerr, maxref, bad, mbad, smax, (bad == 0 && mbad == 0 && smax < 2e-3) ? "OK" : "FAIL");
        fflush(stdout);
        hipFree(din); hipFree(dw); hipFree(dwkw); hipFree(dout); hipFree(dout_old); hipFree(dss); hipFree(dss2); hipFree(dpart); hipFree(dpart_old); hipFree(dzero);
        if (din2) hipFree(din2); if (dmout) hipFree(dmout);
    }
    return 0;
}
