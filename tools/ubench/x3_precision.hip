// Precision + rate probe of the error-corrected split-fp16 product (hi/lo operands, 3 MFMAs per f32 product block) against
// v_mfma_f32_16x16x4_f32 and a float64 host evaluation.  Answers, on the hardware:
//   (1) does ONE f32 accumulator for  hi*hi + hi*lo + lo*hi  keep f32-class accuracy (i.e. how does the MFMA sum internally)?
//   (2) what does leaving lo unscaled (fp16 subnormal lo for |x| < 2^-3) cost, and what does the 4th product (lo*lo) buy?
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/x3_precision.hip -o tools/ubench/x3_precision
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// W [tiles][16][K], X [tiles][16][K] (both "row = output index, K contiguous"), out [tiles][variant][16][16] (row = w index, col = x index)
// variants: 0 f32 MFMA; 1 x3 one accumulator, operands pre-scaled (sw, sx powers of two); 2 x3 unscaled; 3 x4 (with lo*lo) scaled;
//           4 x3 with two accumulators (main, cross) scaled; 5 plain fp16 (hi only)
__global__ void __launch_bounds__(64) probe(const float* __restrict__ W, const float* __restrict__ X, int K, float sw, float sx, float* __restrict__ out) {
    const int lane = threadIdx.x, a = lane & 15, b = lane >> 4;
    const float* w = W + ((size_t)blockIdx.x * 16 + a) * K;
    const float* x = X + ((size_t)blockIdx.x * 16 + a) * K;
    f32x4 acc[6];
    for (int v = 0; v < 6; ++v) acc[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 cross = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 32) {
        // f32: 8 MFMAs of k = 4 (lane holds k = k0 + 4j + b)
        for (int j = 0; j < 8; ++j) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k0 + 4 * j + b], x[k0 + 4 * j + b], acc[0], 0, 0, 0);
        h16x8 wh, wl, xh, xl, whu, wlu, xhu, xlu;
        for (int e = 0; e < 8; ++e) {
            const float wf = w[k0 + 8 * b + e], xf = x[k0 + 8 * b + e];
            const float ws = wf * sw, xs = xf * sx;
            wh[e] = (_Float16)ws; wl[e] = (_Float16)(ws - (float)wh[e]);
            xh[e] = (_Float16)xs; xl[e] = (_Float16)(xs - (float)xh[e]);
            whu[e] = (_Float16)wf; wlu[e] = (_Float16)(wf - (float)whu[e]);
            xhu[e] = (_Float16)xf; xlu[e] = (_Float16)(xf - (float)xhu[e]);
        }
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wlu, xhu, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whu, xlu, acc[2], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whu, xhu, acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xl, acc[3], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, acc[3], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, acc[3], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[3], 0, 0, 0);
        acc[4] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh, acc[4], 0, 0, 0);
        cross = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh, cross, 0, 0, 0);
        cross = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl, cross, 0, 0, 0);
        acc[5] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whu, xhu, acc[5], 0, 0, 0);
    }
    acc[4] += cross;
    const float inv = 1.0f / (sw * sx);
    acc[1] *= inv; acc[3] *= inv; acc[4] *= inv;
    for (int v = 0; v < 6; ++v)
        for (int r = 0; r < 4; ++r) out[(((size_t)blockIdx.x * 6 + v) * 16 + 4 * b + r) * 16 + a] = acc[v][r];
}

// Rate: NM back-to-back MFMAs per wave on NACC independent accumulators, 4 waves per workgroup, 1024 workgroups
template <int NACC>
__global__ void __launch_bounds__(256) rate(float* out, int iters) {
    h16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static double urand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (double)((rs >> 11) & ((1ull << 53) - 1)) / 9007199254740992.0; }
static double nrand() { const double u = urand() + 1e-300, v = urand(); return sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v); }

int main() {
    const int tiles = 64;
    const char* vn[6] = {"f32 mfma 16x16x4      ", "x3 1 acc, scaled      ", "x3 1 acc, unscaled    ", "x4 (with lo*lo) scaled", "x3 2 acc, scaled      ", "fp16 hi only          "};
    struct Cfg { const char* name; int K; double wstd, xstd; bool relu; float sw, sx; };
    const Cfg cfgs[] = {
        {"3x3 256ch  w~N(0,.03) x=relu(N(0,1))      ", 2304, 0.03, 1.0, true, 16384.f, 256.f},
        {"1x1 2048ch w~N(0,.02) x=relu(N(0,1))      ", 2048, 0.02, 1.0, true, 16384.f, 256.f},
        {"1x1 64ch   w~N(0,.1)  x=relu(N(0,1))      ", 64, 0.1, 1.0, true, 4096.f, 256.f},
        {"small act  w~N(0,.03) x=relu(N(0,.01))    ", 2304, 0.03, 0.01, true, 16384.f, 256.f},
        {"DT  d=512  w~N(0,.05) x=N(0,1)            ", 512, 0.05, 1.0, false, 8192.f, 256.f},
    };
    for (const Cfg& c : cfgs) {
        const int K = c.K;
        std::vector<float> W((size_t)tiles * 16 * K), X((size_t)tiles * 16 * K);
        for (auto& v : W) v = (float)(c.wstd * nrand());
        for (auto& v : X) { double t = c.xstd * nrand(); if (c.relu && t < 0) t = 0; v = (float)t; }
        float *dW, *dX, *dO;
        CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dO, (size_t)tiles * 6 * 256 * 4));
        CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(probe, dim3(tiles), dim3(64), 0, 0, dW, dX, K, c.sw, c.sx, dO);
        CK(hipDeviceSynchronize());
        std::vector<float> O((size_t)tiles * 6 * 256);
        CK(hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost));
        double se[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0}, sref = 0; size_t cnt = 0;
        for (int t = 0; t < tiles; ++t)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double r = 0, mag = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)W[((size_t)t * 16 + i) * K + k] * (double)X[((size_t)t * 16 + j) * K + k]; r += p; mag += fabs(p); }
                    sref += mag; ++cnt;
                    for (int v = 0; v < 6; ++v) {
                        const double e = fabs((double)O[(((size_t)t * 6 + v) * 16 + i) * 16 + j] - r) / mag;   // error relative to sum |a b|
                        se[v] += e * e; if (e > mx[v]) mx[v] = e;
                    }
                }
        printf("%s K=%d  (errors relative to sum|w x|; f32 eps 2^-24 = 6.0e-8)\n", c.name, K);
        for (int v = 0; v < 6; ++v) printf("    %s rms %.3e  max %.3e\n", vn[v], sqrt(se[v] / cnt), mx[v]);
        hipFree(dW); hipFree(dX); hipFree(dO);
    }
    // rate
    float* dR; CK(hipMalloc(&dR, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((rate<12>), dim3(1024), dim3(256), 0, 0, dR, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 1024.0 * 4 * iters * 12 * 2.0 * 16 * 16 * 32;
        printf("rate 16x16x32 f16, 12 accumulators: %.3f ms -> %.0f TFLOP/s raw = %.0f TFLOP/s of f32-equivalent products at 3 MFMAs each\n", ms, fl / ms / 1e9, fl / ms / 1e9 / 3);
    }
    return 0;
}
