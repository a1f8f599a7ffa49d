// msda_window.hip -- LDS-staged sampling-window MSDA kernels (placeholder until implemented).
#include "msda_internal.h"

namespace msda {
bool window_supports(const Problem &, bool) { return false; }
void launch_window_forward(const Problem &) {}
void launch_window_backward(const Problem &) {}
}  // namespace msda
