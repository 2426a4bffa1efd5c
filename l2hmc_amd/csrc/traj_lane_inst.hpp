// instantiation helpers of traj_lane.hpp: one translation unit per energy kind (compile time)
#pragma once
#include "traj_lane.hpp"

namespace l2hmc {

int launch_lane_a(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_b(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_c2(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_c3(const KArgs& k, const float* wx, const float* wv, hipStream_t s);

template <int EK, int DP>
int launch_lane_dp(const KArgs& k, const float* wx, const float* wv, hipStream_t s) {
  const unsigned blocks = (unsigned)((k.N + 63) / 64);
  if (lane_hu(k.H) == 10) hipLaunchKernelGGL((traj_lane_kernel<EK, DP, 5>), dim3(blocks), dim3(64), 0, s, k, wx, wv, k.masks, k.trig, k.mu, k.prec, k.logc);
  else hipLaunchKernelGGL((traj_lane_kernel<EK, DP, 8>), dim3(blocks), dim3(64), 0, s, k, wx, wv, k.masks, k.trig, k.mu, k.prec, k.logc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(L2HMC_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
  return L2HMC_OK;
}

// the dimension counts compiled: d <= 4 (measured, tools/bench_lane.py: from d = 8 on the scalar-load latency of a
// net's 2 x (2 d H + ...) weights -- 21 KB at d = 50, more than the 16 KB scalar cache -- makes this form lose to the tiles)
#define L2HMC_LANE_DEFINE(EKc, tag)                                                              \
  int launch_lane_##tag(const KArgs& k, const float* wx, const float* wv, hipStream_t s) {        \
    const int d = k.d;                                                                            \
    if (d <= 2) return launch_lane_dp<EKc, 2>(k, wx, wv, s);                                      \
    if (d <= 4) return launch_lane_dp<EKc, 4>(k, wx, wv, s);                                      \
    return fail(L2HMC_ERR_UNSUPPORTED, "lane kernel: d = %s%lld", "", (long long)d);              \
  }
#define L2HMC_LANE_DEFINE_SMALL(EKc, tag, DMAX) L2HMC_LANE_DEFINE(EKc, tag)

}  // namespace l2hmc
