// Host-side helper: run a block exactly once per (call site, device), thread-safe.  Used for
// hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute is a property of the kernel ON A DEVICE, so a process
// that drives a second GPU (or two host threads racing to the first launch) must not skip it -- the launch of a kernel
// that needs more than 64 KB of dynamic LDS would fail there.  std::call_once blocks concurrent callers until the block
// has run, so no thread launches before the attribute is set.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>

#define RLIPV2_ONCE_PER_DEVICE(...)                                                                                   \
    do {                                                                                                              \
        static std::once_flag once_flags_[64];                                                                        \
        int once_dev_ = 0;                                                                                            \
        (void)hipGetDevice(&once_dev_);                                                                               \
        std::call_once(once_flags_[once_dev_ & 63], [&] { __VA_ARGS__; });                                           \
    } while (0)
