// Unit of libbusca_hip.so (busca_internal.hpp lists them): the few instantiations of the fused Decision-Transformer kernel that crash hipcc's
// 'AMDGPU Rewrite AGPR-Copy-MFMA' pass, compiled without -mllvm -amdgpu-mfma-vgpr-form (busca_amd/build.py; list: BUSCA_DT_AUX_INSTANCES in
// dt_kernel.hip.inc).  The unit that launches them (busca_dt_f16.hip) declares them `extern template` and launches them like every other instantiation.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdint>

#define BUSCA_DT_AUX_TU 1
#include "dt_kernel.hip.inc"

#define BUSCA_DT_AUX_DEF(PR, M, DD, F, EE, NC, NT, SP) template __global__ void dt_fused_kernel<PR, M, DD, F, EE, NC, NT, SP>(const DTParams);
BUSCA_DT_AUX_INSTANCES(BUSCA_DT_AUX_DEF)
