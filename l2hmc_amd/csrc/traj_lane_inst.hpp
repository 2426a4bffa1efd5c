// instantiation helpers of traj_lane.hpp: one translation unit per energy kind (compile time)
#pragma once
#include <cstdlib>

#include "traj_lane.hpp"

namespace l2hmc {

int launch_lane_a(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_b(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_c2(const KArgs& k, const float* wx, const float* wv, hipStream_t s);
int launch_lane_c3(const KArgs& k, const float* wx, const float* wv, hipStream_t s);

// where the d <= 2, H <= 10 kernels keep their weights (traj_lane.hpp, RES; measured: profiles/r06_lane_resident.txt, results are
// bit-identical in all three forms): XNet's layer 2 and heads as VGPR pairs (1) while the chip holds at most two waves per SIMD
// -- x 1.11-1.19 from 16 384 to 131 072 chains --, scalar loads in the loop (0) beyond that (four waves per SIMD hide them, the
// resident forms cap the occupancy at two); the diagonal Gaussian, whose grad U re-reads its parameters by scalar loads inside
// every evaluation, takes all weights in VGPRs, four per register (2): x 2.4-3.5 at every chain count.  L2HMC_LANE_RES=0/1/2 in
// the environment overrides the choice (A/B runs, tools/bench_lane_resident.py; tests).
inline int lane_resident(int ekind, long long n_chains) {
  const char* e = getenv("L2HMC_LANE_RES");
  if (e != nullptr && e[0] >= '0' && e[0] <= '2') return e[0] - '0';
  if (ekind == L2HMC_ENERGY_GAUSS_DIAG) return 2;
  return n_chains < 3LL * 64 * 1024 ? 1 : 0;          // (1024 SIMDs on gfx950: three waves each)
}

template <int EK, int DP>
int launch_lane_dp(const KArgs& k, const float* wx, const float* wv, hipStream_t s) {
  const unsigned blocks = (unsigned)((k.N + 63) / 64);
  const int res = (DP == 2 && lane_hu(k.H) == 10) ? lane_resident(EK, k.N) : 0;
  if (res == 2)
    hipLaunchKernelGGL((traj_lane_kernel<EK, DP, 5, DP == 2 ? 2 : 0>), dim3(blocks), dim3(64), 0, s, k, wx, wv, k.masks, k.trig, k.mu, k.prec, k.logc);
  else if (res == 1)
    hipLaunchKernelGGL((traj_lane_kernel<EK, DP, 5, DP == 2 ? 1 : 0>), dim3(blocks), dim3(64), 0, s, k, wx, wv, k.masks, k.trig, k.mu, k.prec, k.logc);
  else if (lane_hu(k.H) == 10) hipLaunchKernelGGL((traj_lane_kernel<EK, DP, 5>), dim3(blocks), dim3(64), 0, s, k, wx, wv, k.masks, k.trig, k.mu, k.prec, k.logc);
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
