// Fused L2HMC kernels specialised for energy kind 2 (gauss_dense); see l2hmc_kernels.hpp.
#include "traj_small.hpp"

namespace l2hmc {
#define L2HMC_CALL_TRAJ_2(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_kernel<2, DTc, NWc, 3>, k, NWc, lds, s);       \
  else return launch(traj_kernel<2, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_FAST_2(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_fast_kernel<2, DTc, NWc, 3>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<2, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_SMALL_2                                                        \
  if (KH <= 3) return launch(traj_small_kernel<2, 3>, k, 1, lds, s);              \
  else return launch(traj_small_kernel<2, 4>, k, 1, lds, s);
#define L2HMC_CALL_SMALL16_2                                                      \
  if (KH <= 3) return launch(traj_small_kernel<2, 3, 1>, k, 1, lds, s);           \
  else return launch(traj_small_kernel<2, 4, 1>, k, 1, lds, s);
#define L2HMC_CALL_EN_2(DTc, NWc) return launch(energy_kernel<2, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_2(DTc, NWc) return launch(paccept_kernel<2, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(2)
}  // namespace l2hmc
