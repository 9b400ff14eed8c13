// wide_hint.hpp -- should this call launch clahe_interp16_mid_kernel?  (kernels/clahe16.hip.h WideHint; the host half of the hint)
//
// Two words of pinned host memory are stamped by kernels with call sequence numbers: `seen` = the last call in which a rectangle of
// the mid kernel's kind was met (clahe_interp16_kernel), `executed` = the last call whose LUT kernel has run (every call has one).
// The host enqueues ahead of the device -- twenty calls, if the caller likes -- so "lately" is counted in the DEVICE's progress:
// launch while executed - seen <= window.  The difference is taken as a SIGNED number: the host reads the two words at no particular
// moment, the device stamps them at different moments of a call, and a caller's next calls are already numbered while these run -- so
// a reader can find `seen` AHEAD of `executed`; read as unsigned that was 4 billion calls ago and switched the kernel off in the
// middle of a 14-bit stream (found on the GPU in round 6).  Sequence numbers wrap; the signed difference does not care.
// Stand-alone on purpose (no HIP header): tests/cxx/test_host_helpers.cpp.
#ifndef MI_WIDE_HINT_HPP_
#define MI_WIDE_HINT_HPP_
#include <cstdint>

namespace mi_host {

// mode: the option "clahe16_wide" (0 never, 1 by the hint, 2 always)
inline bool mid_kernel_wanted(int mode, uint32_t executed, uint32_t seen, int32_t window)
{
    if (mode >= 2) return true;
    if (mode <= 0) return false;
    return (int32_t)(executed - seen) <= window;
}

}  // namespace mi_host
#endif
