// Fused L2HMC kernels specialised for energy kind 4 (roughwell); see l2hmc_kernels.hpp.
#include "traj_small.hpp"
#include "traj_fast.hpp"

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
#define L2HMC_CALL_SMALL16_4                                                      \
  if (KH <= 3) return launch(traj_small_kernel<4, 3, 1>, k, 1, lds, s);           \
  else return launch(traj_small_kernel<4, 4, 1>, k, 1, lds, s);
#define L2HMC_CALL_EN_4(DTc, NWc) return launch(energy_kernel<4, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_4(DTc, NWc) return launch(paccept_kernel<4, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(4)

}  // namespace l2hmc
