// Stand-alone A/B of the loader-wave persistent 1x1 conv (x3_lw_kernel, busca_amd/csrc/reid_x3p.hip.inc) against the one-shot kernels it replaces
// (conv_x3_kernel<8, 1, 2, 8, X3_PLAIN / X3_BN / X3_MRG, 1, X3_RAW>): same inputs, outputs / statistics / merged block output compared BIT FOR BIT, both timed.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form tools/ubench/x3_lw_bench.hip -o tools/ubench/x3_lw_bench
// Run (GPU box):  tools/ubench/x3_lw_bench [crops]
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
#include "x3_lw_kernel.hip.inc"

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


template <int STG> static void launch_old(const X3Args& a, hipStream_t s) {
    constexpr int tables = STG == X3_BN ? 1 : STG == X3_MRG ? 2 : 0;
    const size_t lds = x3_lds_bytes<8, 1, 2, 8>(a.Cin, tables);
    static bool cfg = false;
    if (!cfg) { CK(hipFuncSetAttribute((const void*)conv_x3_kernel<8, 1, 2, 8, STG, 1, X3_RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<8, 1, 2, 8>(2048, tables))); cfg = true; }
    const unsigned nb = (unsigned)(((a.gridM + 7) / 8) * 8 * a.gridN);
    hipLaunchKernelGGL((conv_x3_kernel<8, 1, 2, 8, STG, 1, X3_RAW>), dim3(nb), dim3(512), lds, s, a);
}
template <int STG, int CT, int D> static void launch_new(X3Args a, hipStream_t s, int ncu, float* dump) {
    static bool cfg = false;
    if (!cfg) { CK(hipFuncSetAttribute((const void*)x3_lw_kernel<STG, CT, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lw_lds_bytes(STG, 2048))); cfg = true; }
    a.gridN = a.Cout / (64 * CT);
    const int per = 8 * a.gridN, nwg = (ncu / per) * per;
    hipLaunchKernelGGL((x3_lw_kernel<STG, CT, D>), dim3(nwg), dim3(512), x3_lw_lds_bytes(STG, a.Cin), s, a, dump);
}

struct Case { const char* name; int ohw, Cin, Cout, stg; bool dss, wts; };

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    int ncu = 256; { hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); ncu = p.multiProcessorCount; }
    const Case cases[] = {
        {"L3 conv1 1024->256, input = previous tail (MRG)  ", 192, 1024, 256, X3_MRG, false, false},
        {"L3 conv1 1024->256 MRG, downsample identity, wts ", 192, 1024, 256, X3_MRG, true, true},
        {"L3 conv3 256->1024 (BN input)                    ", 192, 256, 1024, X3_BN, false, false},
        {"L3 conv3 256->1024 (BN input), weighted          ", 192, 256, 1024, X3_BN, false, true},
        {"L3 b0 conv1 512->256 (plain input, 768 px/crop)  ", 768, 512, 256, X3_PLAIN, false, false},
        {"L4 conv3 512->2048 (BN input)                    ", 48, 512, 2048, X3_BN, false, false},
        {"L4 conv1 2048->512 MRG                           ", 48, 2048, 512, X3_MRG, false, false},
        {"L2 conv1 512->128... as 512->256 plain           ", 768, 512, 256, X3_PLAIN, false, true},
    };
    float* dump; CK(hipMalloc(&dump, 65536));
    for (const Case& c : cases) {
        const int M = n * c.ohw, gridM = (M + 127) / 128;
        const size_t nin = (size_t)M * c.Cin, nout = (size_t)M * c.Cout;
        std::vector<_Float16> hw; std::vector<float> hinv, hss(2 * c.Cin), hssd(2 * c.Cin), hwts(n);
        pack_1x1(c.Cin, c.Cout, hw, hinv);
        for (int i = 0; i < c.Cin; ++i) { hss[2 * i] = 0.8f + 0.4f * frand(); hss[2 * i + 1] = 0.3f * frand(); hssd[2 * i] = 0.9f + 0.3f * frand(); hssd[2 * i + 1] = 0.2f * frand(); }
        for (int i = 0; i < n; ++i) hwts[i] = (float)(1 + (i * 7) % 5);
        float *din, *didt = nullptr, *dout[2], *dmrg[2] = {nullptr, nullptr}, *dpart[2], *dss, *dssd, *dinv, *dzero, *dwts; _Float16* dw;
        unsigned long long* dcnt;
        CK(hipMalloc(&din, nin * 4));
        for (int k = 0; k < 2; ++k) { CK(hipMalloc(&dout[k], nout * 4)); CK(hipMalloc(&dpart[k], (size_t)gridM * 2 * c.Cout * 4)); }
        if (c.stg == X3_MRG) { CK(hipMalloc(&didt, nin * 4)); for (int k = 0; k < 2; ++k) CK(hipMalloc(&dmrg[k], nin * 4)); }
        CK(hipMalloc(&dss, 8 * c.Cin)); CK(hipMalloc(&dssd, 8 * c.Cin)); CK(hipMalloc(&dinv, 4 * c.Cout));
        CK(hipMalloc(&dw, hw.size() * 2)); CK(hipMalloc(&dzero, 256)); CK(hipMemset(dzero, 0, 256)); CK(hipMalloc(&dwts, 4 * n)); CK(hipMalloc(&dcnt, 8));
        CK(hipMemcpy(dss, hss.data(), 8 * c.Cin, hipMemcpyHostToDevice)); CK(hipMemcpy(dssd, hssd.data(), 8 * c.Cin, hipMemcpyHostToDevice));
        CK(hipMemcpy(dinv, hinv.data(), 4 * c.Cout, hipMemcpyHostToDevice)); CK(hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dwts, hwts.data(), 4 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, din, nin, 1u, 2.0f);
        if (didt) hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, didt, nin, 2u, 2.0f);

        X3Args o{};
        o.in = din; o.in_ss = c.stg == X3_PLAIN ? nullptr : dss; o.w = dw; o.inv = dinv; o.zero = dzero; o.wts = c.wts ? dwts : nullptr;
        o.mrg_idt = didt; o.mrg_idt_ss = c.dss ? dssd : nullptr;
        o.M = M; o.Cin = c.Cin; o.Cout = c.Cout; o.H = 1; o.W = c.ohw; o.OH = 1; o.OW = c.ohw; o.stride = 1; o.pad = 0; o.OHWo = c.ohw;
        o.gridM = gridM; o.gridN = c.Cout / 256;

        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        auto time_it = [&](auto&& fn, int iters) { fn(); CK(hipDeviceSynchronize()); CK(hipGetLastError()); CK(hipEventRecord(e0, 0)); for (int i = 0; i < iters; ++i) fn(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                                                   float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3 / iters; };
        auto differs = [&](const float* x, const float* y, size_t cnt) { CK(hipMemset(dcnt, 0, 8)); hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned*)x, (const unsigned*)y, cnt, dcnt);
                                                                         unsigned long long h = 0; CK(hipMemcpy(&h, dcnt, 8, hipMemcpyDeviceToHost)); return h; };
        const double gb = ((double)nin * 4 * (c.stg == X3_MRG ? 3 : 1) + (double)nout * 4) / 1e9, gf = 2.0 * M * c.Cin * (double)c.Cout * 3 / 1e9;
        printf("%s n=%d M=%d: %.2f GB compulsory, %.0f GFLOP of fp16 MFMA\n", c.name, n, M, gb, gf);
        auto prep = [&](int k) { CK(hipMemset(dout[k], 0xff, nout * 4)); CK(hipMemset(dpart[k], 0xff, (size_t)gridM * 2 * c.Cout * 4)); if (dmrg[k]) CK(hipMemset(dmrg[k], 0xff, nin * 4)); };
        prep(0);
        X3Args a0 = o; a0.out = dout[0]; a0.partials = dpart[0]; a0.mrg_out = dmrg[0];
        const double us_old = time_it([&] { if (c.stg == X3_MRG) launch_old<X3_MRG>(a0, 0); else if (c.stg == X3_BN) launch_old<X3_BN>(a0, 0); else launch_old<X3_PLAIN>(a0, 0); }, 10);
        printf("      one-shot kernel                          %8.1f us  %5.2f TB/s  %6.0f TFLOP/s fp16\n", us_old, gb / us_old * 1e3, gf / us_old * 1e-3);
        X3Args a1 = o; a1.out = dout[1]; a1.partials = dpart[1]; a1.mrg_out = dmrg[1];
        auto run_new = [&](const char* label, auto&& fn) {
            prep(1);
            const double us = time_it(fn, 10);
            const unsigned long long d0 = differs(dout[0], dout[1], nout), d1 = differs(dpart[0], dpart[1], (size_t)gridM * 2 * c.Cout), d2 = dmrg[0] ? differs(dmrg[0], dmrg[1], nin) : 0;
            printf("      %-40s %8.1f us  %5.2f TB/s  %6.0f TFLOP/s  x%.2f   out %s, statistics %s, merged %s\n", label, us, gb / us * 1e3, gf / us * 1e-3, us_old / us, d0 ? "DIFFERS" : "==", d1 ? "DIFFERS" : "==", d2 ? "DIFFERS" : "==");
            if (d0 || d1 || d2) printf("        mismatching words: out %llu, statistics %llu, merged %llu\n", d0, d1, d2);
            fflush(stdout);
        };
        if (c.stg == X3_MRG) {
            run_new("loader waves, 64 ch per compute wave", [&] { launch_new<X3_MRG, 4, 2>(a1, 0, ncu, dump); });
            run_new("loader waves, 32 ch per compute wave", [&] { launch_new<X3_MRG, 2, 2>(a1, 0, ncu, dump); });
        } else if (c.stg == X3_BN) {
            run_new("loader waves, 64 ch per compute wave", [&] { launch_new<X3_BN, 4, 4>(a1, 0, ncu, dump); });
            run_new("loader waves, 32 ch per compute wave", [&] { launch_new<X3_BN, 2, 4>(a1, 0, ncu, dump); });
        } else {
            run_new("loader waves, 64 ch per compute wave", [&] { launch_new<X3_PLAIN, 4, 4>(a1, 0, ncu, dump); });
            run_new("loader waves, 32 ch per compute wave", [&] { launch_new<X3_PLAIN, 2, 4>(a1, 0, ncu, dump); });
        }
        hipFree(din); if (didt) hipFree(didt); for (int k = 0; k < 2; ++k) { hipFree(dout[k]); hipFree(dpart[k]); if (dmrg[k]) hipFree(dmrg[k]); }
        hipFree(dss); hipFree(dssd); hipFree(dinv); hipFree(dw); hipFree(dzero); hipFree(dwts); hipFree(dcnt);
    }
    return 0;
}
