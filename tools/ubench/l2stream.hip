// Microbenchmark: every workgroup (one per CU) streams the SAME L2-resident buffer with 1 KiB-per-wave
// global_load_dwordx4, DEPTH loads in flight per wave.  Prints achieved bytes/clk/CU (assuming 2.4 GHz) and GB/s/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ buf, size_t n16_per_wave_pass, int passes, u32x4* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const u32x4* base = buf + (size_t)wave * n16_per_wave_pass + lane;   // each wave its own quarter, like the DT weight slices
    for (int p = 0; p < passes; ++p) {
        for (size_t i = 0; i < n16_per_wave_pass; i += 64 * DEPTH) {
            u32x4 v[DEPTH];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) v[d] = base[i + d * 64];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) acc ^= v[d];
        }
    }
    if (acc[0] == 0x12345678) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int DEPTH> void run(const u32x4* buf, size_t bytes, int grid, u32x4* out) {
    const size_t n16 = bytes / 16 / 4;   // per wave
    const int passes = 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(256), 0, 0, buf, n16, 1, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(stream_kernel<DEPTH>, dim3(grid), dim3(256), 0, 0, buf, n16, passes, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double per_cu = (double)bytes * passes / (ms * 1e-3);
    printf("grid %4d depth %2d buf %5.1f MB: %.3f ms  %.1f GB/s per CU  %.1f B/clk/CU @2.4GHz  aggregate %.2f TB/s\n", grid, DEPTH,
           bytes / 1e6, ms, per_cu / 1e9, per_cu / 2.4e9, per_cu * grid / 1e12);
}

int main() {
    const size_t maxb = 64 << 20;
    u32x4 *buf, *out;
    hipMalloc(&buf, maxb); hipMemset(buf, 1, maxb); hipMalloc(&out, 1 << 22);
    for (size_t mb : {2, 4, 8}) {
        for (int grid : {32, 256}) {
            run<4>(buf, mb << 20, grid, out);
            run<8>(buf, mb << 20, grid, out);
            run<16>(buf, mb << 20, grid, out);
            run<32>(buf, mb << 20, grid, out);
        }
    }
    return 0;
}
