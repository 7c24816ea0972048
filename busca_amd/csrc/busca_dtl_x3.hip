// libbusca_hip.so, layer-wise Decision-Transformer path, split-fp16 (float32-equivalent) flavour (units: busca_internal.hpp).  gfx950 only.
// The two fused layer kernels (QKV projection + attention, out-projection + feed-forward block) run their GEMMs as three fp16 MFMAs per product block on
// hi / lo operands; embed, decoder and the geometries those kernels are not built for run the exact f32 kernels (dt_tiled_host.hpp, GP).
#include "busca_internal.hpp"

#include "dt_kernel.hip.inc"
#include "dt_tiled.hip.inc"
#include "dt_tiled_host.hpp"

int dt_tiled_x3(busca_ctx* c, const DTParams& K, int d, hipStream_t s) {
    if (d == 256) return dt_forward_tiled<2, 256>(c, K, s);
    if (d == 512) return dt_forward_tiled<2, 512>(c, K, s);
    return BUSCA_ENOKERNEL;          // (d = 64: the exact f32 layer-wise path)
}
