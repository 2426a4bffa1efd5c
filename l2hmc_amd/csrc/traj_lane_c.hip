// one-chain-per-lane kernels (traj_lane.hpp): dense Gaussian and mixtures (d <= 16 / d <= 4), and the dispatcher
#include "traj_lane_inst.hpp"
namespace l2hmc {
L2HMC_LANE_DEFINE_SMALL(2, c2, 16)
L2HMC_LANE_DEFINE_SMALL(3, c3, 4)

bool lane_supported(int ek, int d, int H, int ncomp) {
  if (H < 1 || H > 16) return false;
  switch (ek) {
    case L2HMC_ENERGY_GAUSS_DIAG:
    case L2HMC_ENERGY_ROUGHWELL:
    case L2HMC_ENERGY_GAUSS_DENSE: return d <= 4;
    case L2HMC_ENERGY_GMM: return d <= 4 && ncomp >= 1;
    default: return false;
  }
}

int launch_lane(const KArgs& k, hipStream_t s) {
  const LaneLayout L = lane_layout(k.d, k.H);
  const float* wx = k.packed + 2 * (size_t)net_floats(k.NT);
  const float* wv = wx + L.total;
  switch (k.ekind) {
    case L2HMC_ENERGY_GAUSS_DIAG: return launch_lane_a(k, wx, wv, s);
    case L2HMC_ENERGY_ROUGHWELL: return launch_lane_b(k, wx, wv, s);
    case L2HMC_ENERGY_GAUSS_DENSE: return launch_lane_c2(k, wx, wv, s);
    case L2HMC_ENERGY_GMM: return launch_lane_c3(k, wx, wv, s);
    default: return fail(L2HMC_ERR_UNSUPPORTED, "lane kernel: energy kind %s%lld", "", (long long)k.ekind);
  }
}
}  // namespace l2hmc
