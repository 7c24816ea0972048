// libbusca_hip.so, fused Decision-Transformer kernel, fp16-operand flavour (v_mfma_f32_16x16x32_f16) (units: busca_internal.hpp).  gfx950 only.
#include "busca_internal.hpp"

#include "dt_kernel.hip.inc"
#include "dt_launch.hpp"

// (MT token tiles, width d) -> the instantiation built for it; BUSCA_ENOKERNEL: none (busca_dt_forward then runs the layer-wise path)
int dt_fused_f16(busca_ctx* c, const DTParams& K, int MT, int d, bool two, hipStream_t s) {
#define DT_CASE2(M, DD, NCH) if (two && MT == M && d == DD) return dt_launch<1, M, DD, 2 * DD, NCH, 2>(c, K, s)
    DT_CASE2(3, 256, 1); DT_CASE2(2, 256, 1);        // d = 512: the parked f32 residual does not fit the LDS plan
#undef DT_CASE2
#define DT_CASE(M, DD, NCH) if (MT == M && d == DD) return dt_launch<1, M, DD, 2 * DD, NCH>(c, K, s)
    DT_CASE(1, 64, 1); DT_CASE(1, 256, 1); DT_CASE(1, 512, 1); DT_CASE(2, 64, 1); DT_CASE(3, 64, 1); DT_CASE(4, 64, 1); DT_CASE(2, 256, 1); DT_CASE(3, 256, 1); DT_CASE(4, 256, 1); DT_CASE(5, 256, 1); DT_CASE(2, 512, 1); DT_CASE(3, 512, 1); DT_CASE(4, 512, 2);
#undef DT_CASE
    return BUSCA_ENOKERNEL;
}
