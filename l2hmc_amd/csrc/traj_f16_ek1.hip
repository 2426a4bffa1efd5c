// traj_fast_kernel<1, ., ., ., 1>: the f16x2 form of the instruction-lean trajectory kernel for energy kind 1 (gauss_diag);
// its own translation unit so that the build stays parallel.  See traj_fast.hpp.
#include "traj_fast.hpp"

namespace l2hmc {
#define L2HMC_CALL_FAST16(DTc, NWc)                                                 \
  if (KH <= 3) return launch(traj_fast_kernel<1, DTc, NWc, 3, 1>, k, NWc, lds, s);  \
  else return launch(traj_fast_kernel<1, DTc, NWc, 4, 1>, k, NWc, lds, s);
template <>
int launch_fast16_ek<1>(const KArgs& k, int DT, int NW, int KH, long long lds, hipStream_t s) {
  L2HMC_FAST_SWITCH(DT, NW, L2HMC_CALL_FAST16)
}

}  // namespace l2hmc
