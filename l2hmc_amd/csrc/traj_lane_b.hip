// one-chain-per-lane kernels (traj_lane.hpp): Rough Well
#include "traj_lane_inst.hpp"
namespace l2hmc {
L2HMC_LANE_DEFINE(4, b)
}
