// Fused L2HMC kernels specialised for energy kind 5 (funnel); see l2hmc_kernels.hpp.
#include "traj_small.hpp"

namespace l2hmc {
#define L2HMC_CALL_TRAJ_5(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_kernel<5, DTc, NWc, 3>, k, NWc, lds, s);       \
  else return launch(traj_kernel<5, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_FAST_5(DTc, NWc)                                              \
  if (KH <= 3) return launch(traj_fast_kernel<5, DTc, NWc, 3>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<5, DTc, NWc, 4>, k, NWc, lds, s);
#define L2HMC_CALL_SMALL_5 return fail(L2HMC_ERR_UNSUPPORTED, "no small-d kernel for the funnel%s");
#define L2HMC_CALL_SMALL16_5 return fail(L2HMC_ERR_UNSUPPORTED, "no small-d kernel for the funnel%s");
#define L2HMC_CALL_EN_5(DTc, NWc) return launch(energy_kernel<5, DTc, NWc>, k, NWc, lds, s);
#define L2HMC_CALL_PA_5(DTc, NWc) return launch(paccept_kernel<5, DTc, NWc>, k, NWc, lds, s);
L2HMC_DEFINE_LAUNCH_EK(5)
}  // namespace l2hmc
