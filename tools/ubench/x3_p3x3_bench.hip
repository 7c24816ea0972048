// Stand-alone A/B of the persistent halo-resident 3x3 conv of layer 1 (x3_p3x3_kernel, busca_amd/csrc/reid_x3p.hip.inc) against the one-shot ROW3
// kernel (conv_x3_kernel<2, 2, 2, 4, X3_BN, 3, X3_RAW, 0, 3>): same inputs, raw output and per-tile statistics compared BIT FOR BIT, both timed.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/x3_p3x3_bench.hip -o tools/ubench/x3_p3x3_bench
// Run (GPU box):  tools/ubench/x3_p3x3_bench [crops]
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
#include "../../busca_amd/csrc/reid_x3p.hip.inc"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed, float scale) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((float)((z >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f) * scale;
    }
}
__global__ void diff_kernel(const unsigned* a, const unsigned* b, size_t n, unsigned long long* cnt) {
    unsigned long long c = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}

static unsigned long long rs = 0x9E3779B97F4A7C15ull;
static inline float frand() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (float)((rs >> 40) & 0xFFFFFF) / 16777216.0f * 2.f - 1.f; }


int main(int argc, char** argv) {
    int ncu = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    for (int ai = 1; ai < (argc > 1 ? argc : 2); ++ai) {
        const int n = argc > 1 ? atoi(argv[ai]) : 512;
        for (int weighted = 0; weighted < 2; ++weighted) {
            const int OH = 96, OW = 32, C = 64, M = n * OH * OW, ntiles = M / 128, K = 9 * C, nhalf = 18;
            std::vector<float> hw((size_t)C * K), hinv(C), hss(2 * C), hwts(n);
            for (auto& v : hw) v = frand() * 1.7f / sqrtf((float)K);
            for (int co = 0; co < C; co += 7) for (int k = 0; k < K; ++k) hw[(size_t)co * K + k] *= 37.0f;
            std::vector<_Float16> hx3((size_t)C * nhalf * 64);
            for (int co = 0; co < C; ++co) {
                float m = 0.f;
                for (int k = 0; k < K; ++k) m = std::max(m, std::fabs(hw[(size_t)co * K + k]));
                int ex = 0, kc = 0;
                if (m > 0.f) { std::frexp(m, &ex); kc = 13 - ex; }
                hinv[co] = std::ldexp(1.0f, -kc) / X3_XS;
                const int ct = co / 16, a = co % 16;
                for (int h = 0; h < nhalf; ++h)
                    for (int b = 0; b < 4; ++b)
                        for (int e = 0; e < 8; ++e) {
                            const int tap = h >> 1, kk = h & 1;
                            const float ws = std::ldexp(hw[((size_t)co * 9 + tap) * C + kk * 32 + 8 * b + e], kc);
                            const _Float16 hi = (_Float16)ws, lo = (_Float16)(ws - (float)hi);
                            const size_t base = (((size_t)ct * nhalf + h) * 2) * 512 + (size_t)(16 * b + a) * 8 + e;
                            hx3[base] = hi; hx3[base + 512] = lo;
                        }
            }
            for (int i = 0; i < C; ++i) { hss[2 * i] = 0.8f + 0.4f * frand(); hss[2 * i + 1] = 0.3f * frand(); }
            for (int i = 0; i < n; ++i) hwts[i] = (float)(1 + (i * 7) % 5);
            float *din, *dout[2], *dpart[2], *dss, *dinv, *dzero, *dwts; _Float16* dw; unsigned long long* dcnt;
            const size_t nel = (size_t)M * C;
            CK(hipMalloc(&din, nel * 4)); for (int k = 0; k < 2; ++k) { CK(hipMalloc(&dout[k], nel * 4)); CK(hipMalloc(&dpart[k], (size_t)ntiles * 2 * C * 4)); }
            CK(hipMalloc(&dss, 8 * C)); CK(hipMalloc(&dinv, 4 * C)); CK(hipMalloc(&dw, hx3.size() * 2)); CK(hipMalloc(&dzero, 256)); CK(hipMemset(dzero, 0, 256)); CK(hipMalloc(&dwts, 4 * n)); CK(hipMalloc(&dcnt, 8));
            CK(hipMemcpy(dss, hss.data(), 8 * C, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv, hinv.data(), 4 * C, hipMemcpyHostToDevice));
            CK(hipMemcpy(dw, hx3.data(), hx3.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwts, hwts.data(), 4 * n, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, din, nel, 1u, 2.0f);
            for (int k = 0; k < 2; ++k) { CK(hipMemset(dout[k], 0xff, nel * 4)); CK(hipMemset(dpart[k], 0xff, (size_t)ntiles * 2 * C * 4)); }
            X3Args o{};
            o.in = din; o.in_ss = dss; o.w = dw; o.inv = dinv; o.out = dout[0]; o.partials = dpart[0]; o.zero = dzero; o.wts = weighted ? dwts : nullptr;
            o.M = M; o.Cin = C; o.Cout = C; o.H = OH; o.W = OW; o.OH = OH; o.OW = OW; o.stride = 1; o.pad = 1; o.OHWo = OH * OW; o.gridM = ntiles; o.gridN = 1;
            X3P3Args p{};
            p.in = din; p.in_ss = dss; p.w = dw; p.inv = dinv; p.out = dout[1]; p.partials = dpart[1]; p.wts = weighted ? dwts : nullptr; p.M = M; p.OH = OH; p.OHWo = OH * OW; p.ntiles = ntiles;
            const size_t lds_old = x3_lds_bytes<2, 2, 2, 4>(64, 1, false, true);
            CK(hipFuncSetAttribute((const void*)conv_x3_kernel<2, 2, 2, 4, X3_BN, 3, X3_RAW, 0, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
            CK(hipFuncSetAttribute((const void*)x3_p3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3p3_lds_bytes()));
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto time_it = [&](auto&& fn, int iters) { fn(); CK(hipDeviceSynchronize()); CK(hipGetLastError()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                                                       float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / iters; };
            auto differs = [&](const float* x, const float* y, size_t cnt) { CK(hipMemset(dcnt, 0, 8)); hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned*)x, (const unsigned*)y, cnt, dcnt);
                                                                             unsigned long long h = 0; CK(hipMemcpy(&h, dcnt, 8, hipMemcpyDeviceToHost)); return h; };
            const double gb = 2.0 * nel * 4 / 1e9, gf = 2.0 * M * (double)K * C * 3 / 1e9;
            const double us_old = time_it([&] { hipLaunchKernelGGL((conv_x3_kernel<2, 2, 2, 4, X3_BN, 3, X3_RAW, 0, 3>), dim3(((ntiles + 7) / 8) * 8), dim3(256), lds_old, 0, o); }, 10);
            printf("L1 3x3 64->64 n=%d%s: %.2f GB compulsory, %.0f GFLOP fp16 | one-shot ROW3 %8.1f us (%.0f TFLOP/s)\n", n, weighted ? " weighted" : "", gb, gf, us_old, gf / us_old * 1e-3);
            for (int mult = 1; mult <= 1; ++mult) {
                const int nwg = std::min(ntiles, ncu * mult);
                const double us = time_it([&] { hipLaunchKernelGGL(x3_p3x3_kernel, dim3(nwg), dim3(256), x3p3_lds_bytes(), 0, p); }, 10);
                const unsigned long long d0 = differs(dout[0], dout[1], nel), d1 = differs(dpart[0], dpart[1], (size_t)ntiles * 2 * C);
                printf("      persistent halo kernel, %4d workgroups   %8.1f us (%.0f TFLOP/s, %.2f TB/s)  x%.2f   out %s, statistics %s\n", nwg, us, gf / us * 1e-3, gb / us * 1e3, us_old / us, d0 ? "DIFFERS" : "==", d1 ? "DIFFERS" : "==");
                if (d0 || d1) printf("        mismatching words: out %llu, statistics %llu\n", d0, d1);
            }
            fflush(stdout);
            hipFree(din); for (int k = 0; k < 2; ++k) { hipFree(dout[k]); hipFree(dpart[k]); } hipFree(dss); hipFree(dinv); hipFree(dw); hipFree(dzero); hipFree(dwts); hipFree(dcnt);
        }
    }
    return 0;
}
