// Stand-alone check + timing of dtl_ffn_kernel (busca_amd/csrc/dt_tiled.hip.inc) against a float64 host evaluation.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/dtl_ffn_bench.hip -o tools/ubench/dtl_ffn_bench
#define DTL_FFN_UBENCH
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
#include <string>
#include "../../include/busca_hip.h"
#include "../../busca_amd/csrc/dt_kernel.hip.inc"
#include "../../busca_amd/csrc/reid_kernel.hip.inc"
#include "../../busca_amd/csrc/dt_tiled.hip.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static inline float frand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f; }

static void pack(const std::vector<float>& W, int N, int K, int prec, std::vector<unsigned char>& dst) {     // busca_hip.hip pack_matrix
    const int chunk = prec == 0 ? 16 : 32, sub = chunk / 4;
    dst.assign((size_t)(N / 16) * (K / chunk + DTL_TPAD) * 1024, 0);
    size_t off = 0;
    for (int nt = 0; nt < N / 16; ++nt, off += (size_t)DTL_TPAD * 1024)
        for (int kc = 0; kc < K / chunk; ++kc)
            for (int lane = 0; lane < 64; ++lane) {
                const int a = lane & 15, kb = lane >> 4;
                const float* src = &W[(size_t)(16 * nt + a) * K + kc * chunk + kb * sub];
                if (prec == 0) memcpy(&dst[off], src, 16);
                else { _Float16 h[8]; for (int i = 0; i < 8; ++i) h[i] = (_Float16)src[i]; memcpy(&dst[off], h, 16); }
                off += 16;
            }
}

template <int PREC, int D, bool OUTPROJ = false, int PFV = 0>
static void run(int M, int FF) {
    constexpr int ES = PREC == 0 ? 4 : 2;
    std::vector<float> X((size_t)M * D), W1((size_t)FF * D), W2((size_t)D * FF), b1(FF), b2(D), g(D), be(D), O((size_t)M * D), Wo((size_t)D * D), bo(D), g1(D), be1(D);
    for (auto& v : X) v = frand() * 1.5f;
    for (auto& v : O) v = frand() * 1.5f;
    for (auto& v : Wo) v = frand() / sqrtf((float)D) * 1.7f;
    for (auto& v : bo) v = 0.1f * frand();
    for (auto& v : g1) v = 1.0f + 0.2f * frand();
    for (auto& v : be1) v = 0.1f * frand();
    for (auto& v : W1) v = frand() / sqrtf((float)D) * 1.7f;
    for (auto& v : W2) v = frand() / sqrtf((float)FF) * 1.7f;
    for (auto& v : b1) v = 0.1f * frand();
    for (auto& v : b2) v = 0.1f * frand();
    for (auto& v : g) v = 1.0f + 0.2f * frand();
    for (auto& v : be) v = 0.1f * frand();
    std::vector<unsigned char> p1, p2, po; pack(W1, FF, D, PREC, p1); pack(W2, D, FF, PREC, p2); pack(Wo, D, D, PREC, po);
    std::vector<unsigned char> Oop(O.size() * ES);
    for (size_t i = 0; i < O.size(); ++i) { if (PREC == 0) ((float*)Oop.data())[i] = O[i]; else ((_Float16*)Oop.data())[i] = (_Float16)O[i]; }
    void *dwo, *dO; float *dbo, *dg1, *dbe1;
    CK(hipMalloc(&dwo, po.size())); CK(hipMalloc(&dO, Oop.size())); CK(hipMalloc(&dbo, D * 4)); CK(hipMalloc(&dg1, D * 4)); CK(hipMalloc(&dbe1, D * 4));
    CK(hipMemcpy(dwo, po.data(), po.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dO, Oop.data(), Oop.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dbo, bo.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg1, g1.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbe1, be1.data(), D * 4, hipMemcpyHostToDevice));
    std::vector<_Float16> Xh(X.size()); for (size_t i = 0; i < X.size(); ++i) Xh[i] = (_Float16)X[i];
    float *dX, *db1, *db2, *dg, *dbe; _Float16* dXh; void *dw1, *dw2;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dXh, X.size() * 2)); CK(hipMalloc(&dw1, p1.size())); CK(hipMalloc(&dw2, p2.size()));
    CK(hipMalloc(&db1, FF * 4)); CK(hipMalloc(&db2, D * 4)); CK(hipMalloc(&dg, D * 4)); CK(hipMalloc(&dbe, D * 4));
    CK(hipMemcpy(dXh, Xh.data(), X.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dw1, p1.data(), p1.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dw2, p2.data(), p2.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(db1, b1.data(), FF * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db2, b2.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), D * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbe, be.data(), D * 4, hipMemcpyHostToDevice));
    float* dH; CK(hipMalloc(&dH, (size_t)M * FF * 4)); CK(hipMemset(dH, 0, (size_t)M * FF * 4));
    DTLFfnArgs f{}; f.dbg_h = dH; f.Oop = dO; f.w_out = (const u32x4*)dwo; f.b_out = dbo; f.g1 = dg1; f.be1 = dbe1;
    f.Xop = PREC == 0 ? (const void*)dX : (const void*)dXh; f.X = dX; f.Xh = dXh; f.w1 = (const u32x4*)dw1; f.w2 = (const u32x4*)dw2; f.b1 = db1; f.b2 = db2; f.gamma = dg; f.beta = dbe; f.M = M; f.FF = FF; f.act = 0;
    constexpr int BMF = DTLFfnGeom<PREC, D>::BM;
    const size_t lds = DTLFfnGeom<PREC, D>::LDS;
    CK(hipFuncSetAttribute((const void*)dtl_ffn_kernel<PREC, D, OUTPROJ, PFV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    fprintf(stderr, "launching <%d, %d, %d> M=%d\n", PREC, D, (int)OUTPROJ, M);
    for (int rep = 0; rep < 3; ++rep) {       // the kernel updates X in place: restore it every time
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dXh, Xh.data(), X.size() * 2, hipMemcpyHostToDevice));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((dtl_ffn_kernel<PREC, D, OUTPROJ, PFV>), dim3((M + BMF - 1) / BMF), dim3(64 * DTLFfnGeom<PREC, D>::NWV), lds, 0, f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
#ifdef DTL_FFN_TS
    if (M > 20000) {        // phase breakdown of one more launch
        const int nwg = std::min((M + BMF - 1) / BMF, 1024);
        unsigned long long* dts; CK(hipMalloc(&dts, (size_t)1024 * 8 * 8 * 8)); CK(hipMemset(dts, 0, (size_t)1024 * 8 * 8 * 8));
        f.ts = dts;
        hipLaunchKernelGGL((dtl_ffn_kernel<PREC, D, OUTPROJ, PFV>), dim3((M + BMF - 1) / BMF), dim3(64 * DTLFfnGeom<PREC, D>::NWV), lds, 0, f);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> ts((size_t)1024 * 64); CK(hipMemcpy(ts.data(), dts, ts.size() * 8, hipMemcpyDeviceToHost));
        double ph[8] = {0}; for (int w = 0; w < nwg * DTLFfnGeom<PREC, D>::NWV; ++w) for (int k = 0; k < 8; ++k) ph[k] += (double)ts[(size_t)w * 8 + k];
        double tot = 0; for (int k = 0; k < 8; ++k) tot += ph[k];
        const double nwv = (double)nwg * DTLFfnGeom<PREC, D>::NWV;
        printf("   phases per wave (100 MHz ticks): stage %.0f | out-proj GEMM %.0f | +res, LN1, x1 stores %.0f | FFN1 GEMMs %.0f | act + stores + barrier %.0f | FFN2 GEMMs %.0f | barrier %.0f | +res, LN2, stores %.0f | total %.0f\n",
               ph[0] / nwv, ph[1] / nwv, ph[2] / nwv, ph[3] / nwv, ph[4] / nwv, ph[5] / nwv, ph[6] / nwv, ph[7] / nwv, tot / nwv);
        f.ts = nullptr; hipFree(dts);
    }
#endif
    std::vector<float> out(X.size());
    CK(hipMemcpy(out.data(), dX, X.size() * 4, hipMemcpyDeviceToHost));
    // reference (sampled rows), operands rounded as the kernel rounds them
    auto rnd = [&](double v) { return PREC == 0 ? (double)(float)v : (double)(float)(_Float16)(float)v; };
    std::vector<float> hH((size_t)M * FF); CK(hipMemcpy(hH.data(), dH, hH.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxh = 0; int rows = 0; double hyp[4] = {0, 0, 0, 0};
    for (int m = 0; m < M; m += (M > 64 ? M / 37 : 1)) {
        std::vector<double> h(FF), o(D), x1(D);
        for (int c = 0; c < D; ++c) x1[c] = X[(size_t)m * D + c];
        if (OUTPROJ) {
            double mn = 0;
            for (int c = 0; c < D; ++c) { double s = 0; for (int k = 0; k < D; ++k) s += rnd(O[(size_t)m * D + k]) * rnd(Wo[(size_t)c * D + k]); x1[c] = s + bo[c] + X[(size_t)m * D + c]; mn += x1[c]; }
            mn /= D; double vr = 0; for (int c = 0; c < D; ++c) vr += (x1[c] - mn) * (x1[c] - mn); vr /= D;
            for (int c = 0; c < D; ++c) x1[c] = (double)(float)((x1[c] - mn) / sqrt(vr + 1e-5) * g1[c] + be1[c]);
        }
        for (int j = 0; j < FF; ++j) { double s = 0; for (int k = 0; k < D; ++k) s += rnd(x1[k]) * rnd(W1[(size_t)j * D + k]); s += b1[j]; h[j] = rnd(s > 0 ? s : 0); }
        for (int j = 0; j < FF; ++j) { const double e = fabs(h[j] - hH[(size_t)m * FF + j]); if (e > maxh) maxh = e; if (getenv("DBG") && e > 1e-2 && rows < 1) printf("  h mismatch row %d feature %d: got %f want %f\n", m, j, hH[(size_t)m * FF + j], h[j]); }
        double mean = 0;
        for (int c = 0; c < D; ++c) { double s = 0; for (int j = 0; j < FF; ++j) s += h[j] * rnd(W2[(size_t)c * FF + j]); o[c] = s + b2[c] + x1[c]; mean += o[c]; }
        mean /= D; double var = 0; for (int c = 0; c < D; ++c) var += (o[c] - mean) * (o[c] - mean); var /= D;
        for (int c = 0; c < D; ++c) { const double want = (o[c] - mean) / sqrt(var + 1e-5) * g[c] + be[c]; maxerr = fmax(maxerr, fabs(want - out[(size_t)m * D + c])); }
        if (getenv("DBG") && rows < 2 && PREC == 0 && D == 256) {
            // un-normalise the kernel's row: which pre-LayerNorm columns differ?  (o' = (out - beta) / gamma * std + mean is only defined up to the kernel's own mean / std, so compare centred, scaled rows)
            printf("row %d: pre-LN error by 16-column tile (kernel row re-derived with the reference mean / std):", m);
            for (int t = 0; t < D / 16; ++t) { double e = 0; for (int c = 16 * t; c < 16 * t + 16; ++c) e = fmax(e, fabs(((out[(size_t)m * D + c] - be[c]) / g[c]) * sqrt(var + 1e-5) + mean - o[c])); printf(" %.2f", e); }
            printf("\n");
        }
        // hypotheses for a wrong result: 0 = no FFN term at all, 1 = only hidden block 0, 2 = only hidden block 1, 3 = hidden not activated
        for (int hy = 0; hy < 4 && getenv("DBG") && rows < 2; ++hy) {
            std::vector<double> o2(D); double mn = 0;
            for (int c = 0; c < D; ++c) {
                double sacc = 0;
                for (int j = 0; j < FF; ++j) {
                    if (hy == 0) continue;
                    if (hy == 1 && j >= D) continue;
                    if (hy == 2 && j < D) continue;
                    double hv = h[j];
                    if (hy == 3) { double t = 0; for (int k = 0; k < D; ++k) t += rnd(x1[k]) * rnd(W1[(size_t)j * D + k]); hv = rnd(t + b1[j]); }
                    sacc += hv * rnd(W2[(size_t)c * FF + j]);
                }
                o2[c] = sacc + b2[c] + x1[c]; mn += o2[c];
            }
            mn /= D; double vr = 0; for (int c = 0; c < D; ++c) vr += (o2[c] - mn) * (o2[c] - mn); vr /= D;
            for (int c = 0; c < D; ++c) hyp[hy] = fmax(hyp[hy], fabs((o2[c] - mn) / sqrt(vr + 1e-5) * g[c] + be[c] - out[(size_t)m * D + c]));
            if (hy == 3 && rows > 2) break;
        }
        ++rows;
    }
    const double fl = 4.0 * M * D * (double)FF;
    printf("dtl_ffn_kernel<%d, %d, %d, pf %d> M=%d ff=%d: %.1f us, %.1f TFLOP/s, max |err| %.2e (hidden layer %.2e) over %d sampled rows  %s\n", PREC, D, (int)OUTPROJ, PFV, M, FF, ms * 1e3, (fl + (OUTPROJ ? 2.0 * M * D * (double)D : 0.0)) / ms / 1e9, maxerr, maxh, rows,
           maxerr < (PREC == 0 ? 2e-4 : 2e-2) ? "OK" : "FAIL");
    fflush(stdout);
    if (maxerr >= (PREC == 0 ? 2e-4 : 2e-2)) printf("    distance to: no FFN term %.2e | hidden block 0 only %.2e | hidden block 1 only %.2e | no activation %.2e\n", hyp[0], hyp[1], hyp[2], hyp[3]);
    hipFree(dwo); hipFree(dO); hipFree(dbo); hipFree(dg1); hipFree(dbe1); hipFree(dH); hipFree(dX); hipFree(dXh); hipFree(dw1); hipFree(dw2); hipFree(db1); hipFree(db2); hipFree(dg); hipFree(dbe);
}

int main(int argc, char** argv) {
    if (argc > 1 && atoi(argv[1]) == 1) { run<1, 512, true>(64, 1024); run<1, 512, true>(800, 1024); return 0; }
    if (argc > 1 && atoi(argv[1]) == 2) { run<1, 512, false>(64, 1024); run<1, 512, false>(800, 1024); return 0; }
    if (argc > 1 && atoi(argv[1]) == 3) { run<0, 512, true>(64, 1024); run<0, 512, true>(800, 1024); return 0; }
    run<0, 128>(150, 256); run<1, 128>(150, 256); run<0, 384>(150, 768); run<1, 384>(150, 768);
    run<1, 256>(1504, 512); run<0, 256>(1504, 512); run<0, 256>(64, 256);
    run<1, 256, true>(1504, 512); run<0, 256, true>(1504, 512); run<1, 512, true>(800, 1024); run<0, 512, true>(800, 1024); run<1, 384, true>(150, 768); run<1, 512>(800, 1024); run<0, 512>(800, 1024);
    run<1, 512, false>(800, 1024); run<1, 512, true>(800, 1024); run<0, 512, true>(800, 1024);
    if (getenv("BIG")) { run<1, 512, false, 2>(73216, 1024); run<1, 512, false, 4>(73216, 1024); run<1, 512, true, 2>(73216, 1024); run<1, 512, true, 4>(73216, 1024);
                         run<0, 512>(20224, 1024); run<0, 512, true>(20224, 1024); run<1, 256, true>(73216, 512); }
    return 0;
}
