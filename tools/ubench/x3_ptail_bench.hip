// Stand-alone A/B of the persistent block-tail kernel (busca_amd/csrc/reid_x3p.hip.inc) against the one-shot kernels it replaces
// (conv_x3_kernel<8, 1, 2, 8, X3_BN, 1, X3_MERGE_C1 / X3_MERGE>): same inputs, outputs compared BIT FOR BIT on the device, both timed.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/x3_ptail_bench.hip -o tools/ubench/x3_ptail_bench
// Run (GPU box):  tools/ubench/x3_ptail_bench [crops]
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

// hi / lo fragment order + per-channel descale of a 1x1 conv, as busca_reid_load_weights_ex packs them
static void pack_1x1(int Cin, int Cout, std::vector<_Float16>& hx3, std::vector<float>& hinv) {
    const int cch = Cin / 64, nhalf = 2 * cch;
    std::vector<float> hw((size_t)Cout * Cin);
    const float wsc = 1.0f / sqrtf((float)Cin);
    for (auto& v : hw) v = frand() * wsc * 1.7f;
    for (int co = 0; co < Cout; co += 7) for (int k = 0; k < Cin; ++k) hw[(size_t)co * Cin + k] *= 37.0f;
    hx3.assign((size_t)Cout * nhalf * 64, (_Float16)0.f); hinv.assign(Cout, 0.f);
    for (int co = 0; co < Cout; ++co) {
        float m = 0.f;
        for (int k = 0; k < Cin; ++k) m = std::max(m, std::fabs(hw[(size_t)co * Cin + k]));
        int ex = 0, kc = 0;
        if (m > 0.f) { std::frexp(m, &ex); kc = 13 - ex; }
        hinv[co] = std::ldexp(1.0f, -kc) / X3_XS;
        const int ct = co / 16, a = co % 16;
        for (int h = 0; h < nhalf; ++h)
            for (int b = 0; b < 4; ++b)
                for (int e = 0; e < 8; ++e) {
                    const int chunk = h >> 1, kk = h & 1;
                    const float ws = std::ldexp(hw[(size_t)co * Cin + chunk * 64 + kk * 32 + 8 * b + e], kc);
                    const _Float16 hi = (_Float16)ws, lo = (_Float16)(ws - (float)hi);
                    const size_t base = (((size_t)ct * nhalf + h) * 2) * 512 + (size_t)(16 * b + a) * 8 + e;
                    hx3[base] = hi; hx3[base + 512] = lo;
                }
    }
}

template <int EPI> static void launch_old(const X3Args& a, hipStream_t s) {
    const size_t lds = x3_lds_bytes<8, 1, 2, 8>(a.Cin, 1, EPI == X3_MERGE_C1);
    static bool cfg = false;
    if (!cfg) { CK(hipFuncSetAttribute((const void*)conv_x3_kernel<8, 1, 2, 8, X3_BN, 1, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<8, 1, 2, 8>(2048, 1, EPI == X3_MERGE_C1))); cfg = true; }
    const unsigned nb = (unsigned)(((a.gridM + 7) / 8) * 8 * a.gridN);
    hipLaunchKernelGGL((conv_x3_kernel<8, 1, 2, 8, X3_BN, 1, EPI>), dim3(nb), dim3(512), lds, s, a);
}
template <int KC, int C1, int NW, bool W2P = true, bool W3P = true> static void launch_new(const X3PArgs& a, hipStream_t s, int nwg) {
    static bool cfg = false;
    if (!cfg) { CK(hipFuncSetAttribute((const void*)x3_ptail_kernel<KC, C1, NW, W2P, W3P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3p_lds_bytes<KC, C1>())); cfg = true; }
    hipLaunchKernelGGL((x3_ptail_kernel<KC, C1, NW, W2P, W3P>), dim3(nwg), dim3(64 * NW), (x3p_lds_bytes<KC, C1>()), s, a);
}

struct Case { const char* name; int ohw, Cin, Cout, c1; bool dss, wts; };

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    int ncu = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    const Case cases[] = {
        {"L1 tail 64->256 + conv1 256->64          ", 3072, 64, 256, 64, false, false},
        {"L1 b0 tail (downsample identity) + c1 64 ", 3072, 64, 256, 64, true, false},
        {"L1 tail + conv1 256->64, weighted batch  ", 3072, 64, 256, 64, false, true},
        {"L1 last tail 64->256 + conv1 256->128    ", 3072, 64, 256, 128, false, false},
        {"L2 tail 128->512                         ", 768, 128, 512, 0, false, false},
        {"L2 b0 tail (downsample identity)         ", 768, 128, 512, 0, true, false},
    };
    for (const Case& c : cases) {
        const int M = n * c.ohw, ntiles = M / 128;
        const size_t nin = (size_t)M * c.Cin, nout = (size_t)M * c.Cout, nc1 = (size_t)M * (c.c1 ? c.c1 : 1);
        std::vector<_Float16> hw3, hw1; std::vector<float> hinv3, hinv1, hss(2 * c.Cin), hss3(2 * c.Cout), hssd(2 * c.Cout), hwts(n);
        pack_1x1(c.Cin, c.Cout, hw3, hinv3);
        if (c.c1) pack_1x1(256, c.c1, hw1, hinv1);
        for (int i = 0; i < c.Cin; ++i) { hss[2 * i] = 0.8f + 0.4f * frand(); hss[2 * i + 1] = 0.3f * frand(); }
        for (int i = 0; i < c.Cout; ++i) { hss3[2 * i] = 0.8f + 0.4f * frand(); hss3[2 * i + 1] = 0.3f * frand(); hssd[2 * i] = 0.9f + 0.3f * frand(); hssd[2 * i + 1] = 0.2f * frand(); }
        for (int i = 0; i < n; ++i) hwts[i] = (float)(1 + (i * 7) % 5);
        float *din, *didt, *dout[2], *dc1[2], *dpart[2], *dss, *dss3, *dssd, *dinv3, *dinv1 = nullptr, *dzero, *dwts; _Float16 *dw3, *dw1 = nullptr;
        unsigned long long* dcnt;
        CK(hipMalloc(&din, nin * 4)); CK(hipMalloc(&didt, nout * 4));
        for (int k = 0; k < 2; ++k) { CK(hipMalloc(&dout[k], nout * 4)); CK(hipMalloc(&dc1[k], nc1 * 4)); CK(hipMalloc(&dpart[k], (size_t)ntiles * 2 * 128 * 4)); }
        CK(hipMalloc(&dss, 8 * c.Cin)); CK(hipMalloc(&dss3, 8 * c.Cout)); CK(hipMalloc(&dssd, 8 * c.Cout)); CK(hipMalloc(&dinv3, 4 * c.Cout));
        CK(hipMalloc(&dw3, hw3.size() * 2)); CK(hipMalloc(&dzero, 256)); CK(hipMemset(dzero, 0, 256)); CK(hipMalloc(&dwts, 4 * n)); CK(hipMalloc(&dcnt, 8));
        CK(hipMemcpy(dss, hss.data(), 8 * c.Cin, hipMemcpyHostToDevice)); CK(hipMemcpy(dss3, hss3.data(), 8 * c.Cout, hipMemcpyHostToDevice));
        CK(hipMemcpy(dssd, hssd.data(), 8 * c.Cout, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv3, hinv3.data(), 4 * c.Cout, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw3, hw3.data(), hw3.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dwts, hwts.data(), 4 * n, hipMemcpyHostToDevice));
        if (c.c1) { CK(hipMalloc(&dw1, hw1.size() * 2)); CK(hipMalloc(&dinv1, 4 * c.c1)); CK(hipMemcpy(dw1, hw1.data(), hw1.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dinv1, hinv1.data(), 4 * c.c1, hipMemcpyHostToDevice)); }
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, din, nin, 1u, 2.0f);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, didt, nout, 2u, 2.0f);
        for (int k = 0; k < 2; ++k) { CK(hipMemset(dout[k], 0xff, nout * 4)); CK(hipMemset(dc1[k], 0xff, nc1 * 4)); CK(hipMemset(dpart[k], 0xff, (size_t)ntiles * 2 * 128 * 4)); }

        X3Args o{};
        o.in = din; o.in_ss = dss; o.w = dw3; o.inv = dinv3; o.out = dout[0]; o.partials = dpart[0]; o.zero = dzero; o.wts = c.wts ? dwts : nullptr;
        o.out_ss = dss3; o.idt = didt; o.idt_ss = c.dss ? dssd : nullptr; o.c1_w = dw1; o.c1_inv = dinv1; o.c1_out = dc1[0]; o.c1_cout = c.c1;
        o.M = M; o.Cin = c.Cin; o.Cout = c.Cout; o.H = c.ohw / 32; o.W = 32; o.OH = o.H; o.OW = 32; o.stride = 1; o.pad = 0; o.OHWo = c.ohw;
        o.gridM = ntiles; o.gridN = c.Cout / 256;
        X3PArgs p{};
        p.in = din; p.in_ss = dss; p.w = dw3; p.inv = dinv3; p.out = dout[1]; p.out_ss = dss3; p.idt = didt; p.idt_ss = c.dss ? dssd : nullptr;
        p.c1_w = dw1; p.c1_inv = dinv1; p.c1_out = dc1[1]; p.partials = dpart[1]; p.wts = c.wts ? dwts : nullptr;
        p.M = M; p.Cout = c.Cout; p.OHWo = c.ohw; p.ntiles = ntiles; p.gridN = c.Cout / 256;

        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_it = [&](auto&& fn, int iters) { fn(); CK(hipDeviceSynchronize()); CK(hipGetLastError()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                                                   float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / iters; };
        auto differs = [&](const float* x, const float* y, size_t cnt) { CK(hipMemset(dcnt, 0, 8)); hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned*)x, (const unsigned*)y, cnt, dcnt);
                                                                         unsigned long long h = 0; CK(hipMemcpy(&h, dcnt, 8, hipMemcpyDeviceToHost)); return h; };
        const double gb = ((double)nin * 4 + 2.0 * nout * 4 + (c.c1 ? (double)nc1 * 4 : 0.0)) / 1e9;
        printf("%s n=%d M=%d: %.2f GB compulsory\n", c.name, n, M, gb);
        const double us_old = time_it([&] { if (c.c1) launch_old<X3_MERGE_C1>(o, 0); else launch_old<X3_MERGE>(o, 0); }, 10);
        printf("      one-shot kernel                          %8.1f us  %5.2f TB/s\n", us_old, gb / us_old * 1e3);
        auto run_new = [&](const char* label, auto&& fn) {
            CK(hipMemset(dout[1], 0xff, nout * 4)); CK(hipMemset(dc1[1], 0xff, nc1 * 4)); CK(hipMemset(dpart[1], 0xff, (size_t)ntiles * 2 * 128 * 4));
            const double us = time_it(fn, 10);
            const unsigned long long d0 = differs(dout[0], dout[1], nout), d1 = c.c1 ? differs(dc1[0], dc1[1], nc1) : 0, d2 = c.c1 ? differs(dpart[0], dpart[1], (size_t)ntiles * 2 * c.c1) : 0;
            printf("      %-40s %8.1f us  %5.2f TB/s  x%.2f   out %s, conv1 %s, statistics %s\n", label, us, gb / us * 1e3, us_old / us, d0 ? "DIFFERS" : "==", d1 ? "DIFFERS" : "==", d2 ? "DIFFERS" : "==");
            if (d0 || d1 || d2) printf("        mismatching words: out %llu, conv1 %llu, statistics %llu\n", d0, d1, d2);
            fflush(stdout);
        };
        if (c.c1 == 64) {
            run_new("persistent, 4 waves, 1 per CU", [&] { launch_new<1, 64, 4>(p, 0, ncu); });
            run_new("persistent, 4 waves, w2 per half tile", [&] { launch_new<1, 64, 4, false>(p, 0, ncu); });
        } else if (c.c1 == 128) {
            run_new("persistent, 4 waves, w2 + w3 per half tile", [&] { launch_new<1, 128, 4, false, false>(p, 0, ncu); });
        } else {
            run_new("persistent, 4 waves, 1 per CU", [&] { launch_new<2, 0, 4>(p, 0, ncu); });
            run_new("persistent, 8 waves, 1 per CU", [&] { launch_new<2, 0, 8>(p, 0, ncu); });
            run_new("persistent, 8 waves, 2 x CUs workgroups", [&] { launch_new<2, 0, 8>(p, 0, 2 * ncu); });
        }
        hipFree(din); hipFree(didt); for (int k = 0; k < 2; ++k) { hipFree(dout[k]); hipFree(dc1[k]); hipFree(dpart[k]); }
        hipFree(dss); hipFree(dss3); hipFree(dssd); hipFree(dinv3); hipFree(dw3); hipFree(dzero); hipFree(dwts); hipFree(dcnt); if (dw1) hipFree(dw1); if (dinv1) hipFree(dinv1);
    }
    return 0;
}
