// Probe of ds_read_b64_tr_b16 (gfx950): every lane passes the address of its own 8-byte chunk; prints, per lane and element, WHICH lane's chunk and which
// element of it came back.  Build: hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_probe.hip -o tools/ubench/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int l = threadIdx.x;
    for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (unsigned short)(l * 4 + e);     // chunk of lane l = elements 4 l .. 4 l + 3
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + l * 4));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 512); probe<<<1, 64>>>(d);
    unsigned short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) { const int src = h[l * 4 + e]; printf("  (lane %2d, el %d)", src / 4, src % 4);
            const int g = l & ~15, i = l & 15; if (src / 4 != g + 4 * e + i / 4 || src % 4 != i % 4) ok = 0; }
        printf("\n");
    }
    printf("hypothesis lane i elem j <- chunk of lane (group + 4 j + i / 4), element i %% 4: %s\n", ok ? "HOLDS" : "FAILS");
    return 0;
}
