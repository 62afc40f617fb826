// vote.h -- the two-phase CheckFundamental vote (fast_kernels.hip), declared apart from dr3hip_internal.h
#pragma once
#include "dr3hip_internal.h"

namespace dr3 {
constexpr int VOTE_CHUNK = 16384;      // matches per round: bounds the chi-square scratch at VOTE_CHUNK x padded hypotheses x 8 bytes
// floats of scratch launch_check_fundamental_2p needs for a.n_hyp hypotheses
size_t vote_scratch_floats(int n_hyp, int n);
// the vote in two kernels per round of VOTE_CHUNK matches: chi-square values of every (match, hypothesis) in parallel, then the
// reference's ordered accumulation per hypothesis.
hipError_t launch_check_fundamental_2p(const CheckFArgs& a, float* scratch, hipStream_t s);
}  // namespace dr3
