// Fused L2HMC kernels specialised for energy kind 1 (gauss_diag); see l2hmc_kernels.hpp.
#include "traj_small.hpp"
#include "traj_fast.hpp"

namespace l2hmc {
#define L2HMC_CALL_TRAJ_1(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_kernel<1, DTc, NWc, 3>, k, NWc, lds, s);       \
  else return launch(traj_kernel<1, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_FAST_1(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_fast_kernel<1, DTc, NWc, 3>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<1, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_SMALL_1                                                        \
  if (KH <= 3) return launch(traj_small_kernel<1, 3>, k, 1, lds, s);              \
  else return launch(traj_small_kernel<1, 4>, k, 1, lds, s);
#define L2HMC_CALL_SMALL16_1                                                      \
  if (KH <= 3) return launch(traj_small_kernel<1, 3, 1>, k, 1, lds, s);           \
  else return launch(traj_small_kernel<1, 4, 1>, k, 1, lds, s);
#define L2HMC_CALL_EN_1(DTc, NWc) return launch(energy_kernel<1, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_1(DTc, NWc) return launch(paccept_kernel<1, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(1)

}  // namespace l2hmc
