// Fused L2HMC kernels specialised for energy kind 3 (gmm); see l2hmc_kernels.hpp.
#include "traj_small.hpp"

namespace l2hmc {
#define L2HMC_CALL_TRAJ_3(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_kernel<3, DTc, NWc, 3>, k, NWc, lds, s);       \
  else return launch(traj_kernel<3, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_FAST_3(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_fast_kernel<3, DTc, NWc, 3>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<3, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_SMALL_3                                                        \
  if (KH <= 3) return launch(traj_small_kernel<3, 3>, k, 1, lds, s);              \
  else return launch(traj_small_kernel<3, 4>, k, 1, lds, s);
#define L2HMC_CALL_SMALL16_3                                                      \
  if (KH <= 3) return launch(traj_small_kernel<3, 3, 1>, k, 1, lds, s);           \
  else return launch(traj_small_kernel<3, 4, 1>, k, 1, lds, s);
#define L2HMC_CALL_EN_3(DTc, NWc) return launch(energy_kernel<3, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_3(DTc, NWc) return launch(paccept_kernel<3, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(3)
}  // namespace l2hmc
