// libbusca_hip.so, fused Decision-Transformer kernel, split-fp16 (float32-equivalent) flavour: the f32 kernel's shapes (units: busca_internal.hpp).  gfx950 only.
#include "busca_internal.hpp"

#include "dt_kernel.hip.inc"
#include "dt_launch.hpp"

// (MT token tiles, width d) -> the instantiation built for it; BUSCA_ENOKERNEL: none (busca_dt_forward then runs the layer-wise path)
int dt_fused_x3(busca_ctx* c, const DTParams& K, int MT, int d, hipStream_t s) {
#define DT_CASE(M, DD, NCH) if (MT == M && d == DD) return dt_launch<2, M, DD, 2 * DD, NCH>(c, K, s)
    DT_CASE(1, 64, 1); DT_CASE(1, 256, 1); DT_CASE(1, 512, 1); DT_CASE(2, 64, 1); DT_CASE(3, 64, 1); DT_CASE(4, 64, 1); DT_CASE(2, 256, 1); DT_CASE(3, 256, 1); DT_CASE(2, 512, 2);
#undef DT_CASE
    return BUSCA_ENOKERNEL;
}
