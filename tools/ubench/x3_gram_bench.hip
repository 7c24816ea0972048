// Stand-alone check + timing of x3_gram_kernel / x3_gram_reduce_kernel (busca_amd/csrc/reid_x3.hip.inc) against float64.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/x3_gram_bench.hip -o tools/ubench/x3_gram_bench
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
#include "x3_gram_r4.hip.inc"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static inline float frand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f; }

template <int C>
static void run(int n, int ohw, bool weighted) {
    const int M = n * ohw, ntiles = (M + 127) / 128, nwg = std::min(ntiles, C == 64 ? 512 : 256);
    std::vector<float> X((size_t)M * C), ss(2 * C), wts(n);
    for (auto& v : X) v = frand() * 2.f + 0.3f;
    for (int c = 0; c < C; ++c) { ss[2 * c] = 0.8f + 0.4f * frand(); ss[2 * c + 1] = 0.3f * frand(); }
    for (auto& v : wts) v = (float)(1 + (int)(fabsf(frand()) * 3.99f));
    float *dX, *dss, *dw; double *dpart, *dG;
    CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dss, 8 * C)); CK(hipMalloc(&dw, n * 4)); CK(hipMalloc(&dpart, (size_t)nwg * (C * C + C) * 8)); CK(hipMalloc(&dG, (size_t)(C * C + C) * 8));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dss, ss.data(), 8 * C, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, wts.data(), n * 4, hipMemcpyHostToDevice));
    X3GramArgs a{}; a.x = dX; a.in_ss = dss; a.wts = weighted ? dw : nullptr; a.part = dpart; a.M = M; a.OHW = ohw; a.ntiles = ntiles;
    const size_t lds4 = (size_t)2 * C * (128 * 2 + 16) + (size_t)C * 8 + (size_t)(256 / (C / 8)) * C * 8;
    const size_t lds = (size_t)2 * 128 * 2 * C + (size_t)C * 8;
    CK(hipFuncSetAttribute((const void*)x3_gram_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute((const void*)x3_gram_kernel_r4<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    // round-4 kernel on the same data: time + partials for the bit-for-bit comparison
    double* dpart4; CK(hipMalloc(&dpart4, (size_t)nwg * (C * C + C) * 8));
    float ms4 = 0;
    {
        X3GramArgs a4 = a; a4.part = dpart4;
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(f0));
            hipLaunchKernelGGL((x3_gram_kernel_r4<C>), dim3(nwg), dim3(256), lds4, 0, a4);
            CK(hipEventRecord(f1)); CK(hipEventSynchronize(f1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms4, f0, f1));
        }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((x3_gram_kernel<C>), dim3(nwg), dim3(4 * C), lds, 0, a);
        hipLaunchKernelGGL((x3_gram_reduce_kernel<C>), dim3((C * C + C + 63) / 64), dim3(256), 0, 0, (const double*)dpart, nwg, dG, dG + C * C);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError()); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<double> G(C * C + C);
    CK(hipMemcpy(G.data(), dG, G.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> P0((size_t)nwg * (C * C + C)), P4(P0.size());
    CK(hipMemcpy(P0.data(), dpart, P0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(P4.data(), dpart4, P4.size() * 8, hipMemcpyDeviceToHost));
    const bool same = memcmp(P0.data(), P4.data(), P0.size() * 8) == 0;
    hipFree(dpart4);
    // reference: a few (i, j) entries and a few channel sums over ALL pixels
    double maxrel = 0;
    const int pairs[6][2] = {{0, 0}, {1, 17}, {C - 1, C - 1}, {C / 2, 3}, {5, C - 2}, {33, 34}};
    for (auto& pr : pairs) {
        double g = 0, si = 0;
        for (int m = 0; m < M; ++m) {
            const double w = weighted ? wts[m / ohw] : 1.0;
            const double xi = fmax((double)fmaf(X[(size_t)m * C + pr[0]], ss[2 * pr[0]], ss[2 * pr[0] + 1]), 0.0), xj = fmax((double)fmaf(X[(size_t)m * C + pr[1]], ss[2 * pr[1]], ss[2 * pr[1] + 1]), 0.0);
            g += w * xi * xj; si += w * xi;
        }
        maxrel = fmax(maxrel, fabs(G[(size_t)pr[0] * C + pr[1]] - g) / fabs(g));
        maxrel = fmax(maxrel, fabs(G[(size_t)pr[1] * C + pr[0]] - g) / fabs(g));
        maxrel = fmax(maxrel, fabs(G[(size_t)C * C + pr[0]] - si) / fabs(si));
    }
    printf("x3_gram<%d> n=%d ohw=%d (M=%d)%s: %.1f us gram + reduce (%.2f TB/s), max relative error %.2e  %s | round-4 kernel alone %.1f us, partials %s\n", C, n, ohw, M, weighted ? " weighted" : "", ms * 1e3, (double)M * C * 4 / ms / 1e9, maxrel, maxrel < 2e-6 ? "OK" : "FAIL", ms4 * 1e3, same ? "== (bit for bit)" : "DIFFER");
    fflush(stdout);
    hipFree(dX); hipFree(dss); hipFree(dw); hipFree(dpart); hipFree(dG);
}

int main() {
    run<64>(3, 3072, false); run<64>(5, 3072, true); run<128>(5, 768, true); run<128>(7, 768, false);
    run<64>(512, 3072, false); run<128>(512, 768, false); run<64>(88, 3072, true); run<128>(88, 768, true);
    return 0;
}
