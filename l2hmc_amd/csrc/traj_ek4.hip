// Fused L2HMC kernels specialised for energy kind 4 (roughwell); see l2hmc_kernels.hpp.
#include "traj_small.hpp"
#include "traj_tile.hpp"

namespace l2hmc {
#define L2HMC_CALL_TRAJ_4(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_kernel<4, DTc, NWc, 3>, k, NWc, lds, s);       \
  else return launch(traj_kernel<4, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_FAST_4(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_fast_kernel<4, DTc, NWc, 3>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<4, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_SMALL_4                                                        \
  if (KH <= 3) return launch(traj_small_kernel<4, 3>, k, 1, lds, s);              \
  else return launch(traj_small_kernel<4, 4>, k, 1, lds, s);
#define L2HMC_CALL_EN_4(DTc, NWc) return launch(energy_kernel<4, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_4(DTc, NWc) return launch(paccept_kernel<4, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(4)

template <class K>
static int launch_tile(K kern, const KArgs& k, long long lds, hipStream_t s) {
  if (lds > kMaxLdsBytes) return fail(L2HMC_ERR_UNSUPPORTED, "tile kernel: %s%lld bytes of LDS needed", "", lds);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long long blocks = (k.N + 63) / 64;
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), (size_t)lds, s, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}
template <>
int launch_tile_ek<4>(const KArgs& k, int DT, int KH, long long lds, hipStream_t s) {
  if (DT == 3) return KH <= 3 ? launch_tile(traj_tile_kernel<4, 3, 3, 4>, k, lds, s) : launch_tile(traj_tile_kernel<4, 3, 4, 4>, k, lds, s);
  return KH <= 3 ? launch_tile(traj_tile_kernel<4, 4, 3, 4>, k, lds, s) : launch_tile(traj_tile_kernel<4, 4, 4, 4>, k, lds, s);
}
}  // namespace l2hmc
