// Register-resident whole-rollout kernel; placeholder until the kernel lands.
#include "kernels.hpp"

namespace eamrl {
bool rollout_resident_supports(int, const DecArgs&) { return false; }
int launch_rollout_resident(int, const DecArgs&, hipStream_t) { return EAMRL_E_ARG; }
}  // namespace eamrl
