// What the host-model build of the MSDA library leaves out: msda_window.hip (round 1's sorted-scatter kernels use inline
// DPP-operand assembly that the lane-level model does not cover).  Its variant is reported as unsupported, so "auto" never
// picks it and an explicit request returns MSDA_ERR_BAD_VARIANT.
#define MSDA_EMU 1
#include <hip/hip_runtime.h>

#include "../../rlipv2_amd/csrc/msda_internal.h"

namespace msda {
bool window_supports(const Problem &, bool) { return false; }
void launch_window_forward(const Problem &) { std::abort(); }
void launch_window_backward(const Problem &) { std::abort(); }
}  // namespace msda
