// libbusca_hip.so, layer-wise Decision-Transformer path, exact float32 (units: busca_internal.hpp).  gfx950 only.
#include "busca_internal.hpp"

#include "dt_kernel.hip.inc"
#include "dt_tiled.hip.inc"
#include "dt_tiled_host.hpp"

int dt_tiled_f32(busca_ctx* c, const DTParams& K, int d, hipStream_t s) {
    if (d == 64) return dt_forward_tiled<0, 64>(c, K, s);
    if (d == 256) return dt_forward_tiled<0, 256>(c, K, s);
    if (d == 512) return dt_forward_tiled<0, 512>(c, K, s);
    return fail(c, BUSCA_EINVAL, "no layer-wise Decision-Transformer kernels for d=%d", d);
}
