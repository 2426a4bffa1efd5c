// The one-wave-per-tile kernels (traj_tile.hpp) of both elementwise targets, in a translation unit of their own: the
// instruction-lean 4-wave kernels next door (traj_ek1, traj_ek4) are compiled with LLVM's max-ILP scheduling strategy
// (+1 % on the headline configuration), which costs this kernel registers it does not have (256 VGPRs + scratch, -2 % at
// 16 384 chains: profiles/r03_exchange_variants.txt).
#include "traj_tile.hpp"

namespace l2hmc {

template <class K>
static int launch_tile(K kern, int TPW, const KArgs& k, long long lds, hipStream_t s) {
  if (lds > kMaxLdsBytes) return fail(L2HMC_ERR_UNSUPPORTED, "tile kernel: %s%lld bytes of LDS needed", "", lds);
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  }
  const long long blocks = (k.N + 16 * TPW - 1) / (16 * TPW);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * TPW), (size_t)lds, s, k);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}
template <>
int launch_tile_ek<1>(const KArgs& k, int DT, int KH, int tpw, long long lds, hipStream_t s) {
  const bool half = k.d - 16 * (DT - 1) <= 2;      // the last slice holds <= 2 dimensions: transcendentals on 2 of 4 components
#define L2HMC_TILE_GO(DTc, KHc)                                                                       \
  return half ? (tpw == 8 ? launch_tile(traj_tile_kernel<1, DTc, KHc, 8, true>, 8, k, lds, s) : launch_tile(traj_tile_kernel<1, DTc, KHc, 4, true>, 4, k, lds, s)) \
              : (tpw == 8 ? launch_tile(traj_tile_kernel<1, DTc, KHc, 8, false>, 8, k, lds, s) : launch_tile(traj_tile_kernel<1, DTc, KHc, 4, false>, 4, k, lds, s));
  if (DT == 3) { if (KH <= 3) { L2HMC_TILE_GO(3, 3) } else { L2HMC_TILE_GO(3, 4) } }
  if (KH <= 3) { L2HMC_TILE_GO(4, 3) } else { L2HMC_TILE_GO(4, 4) }
#undef L2HMC_TILE_GO
}
template <>
int launch_tile_ek<4>(const KArgs& k, int DT, int KH, int tpw, long long lds, hipStream_t s) {
  const bool half = k.d - 16 * (DT - 1) <= 2;
#define L2HMC_TILE_GO(DTc, KHc)                                                                       \
  return half ? (tpw == 8 ? launch_tile(traj_tile_kernel<4, DTc, KHc, 8, true>, 8, k, lds, s) : launch_tile(traj_tile_kernel<4, DTc, KHc, 4, true>, 4, k, lds, s)) \
              : (tpw == 8 ? launch_tile(traj_tile_kernel<4, DTc, KHc, 8, false>, 8, k, lds, s) : launch_tile(traj_tile_kernel<4, DTc, KHc, 4, false>, 4, k, lds, s));
  if (DT == 3) { if (KH <= 3) { L2HMC_TILE_GO(3, 3) } else { L2HMC_TILE_GO(3, 4) } }
  if (KH <= 3) { L2HMC_TILE_GO(4, 3) } else { L2HMC_TILE_GO(4, 4) }
#undef L2HMC_TILE_GO
}
}  // namespace l2hmc
