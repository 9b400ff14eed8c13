// mi_lumaeq.hip -- C ABI (include/mi_lumaeq.h) over the gfx950 kernels in lumaeq_kernels.hip.h.
//
// Boundary being replaced (reference file:line):
//   cv::equalizeHist call site            OpenCVequalHist.cpp:145, nextimprovement.cpp:168
//   cv::CLAHE::apply call site            clahevideo.cpp:195, clahe1frame.cpp:93
//   FPGA backend host sequence            OpenCLequalHist.cpp:346-365 (setArg x5, write x2, task, read)
//   per-worker device objects + buffers   OpenCLequalHist.cpp:142-152, :175-186
// There is NO CPU fallback in this file: without a HIP device every entry point fails loudly.
#include "../../include/mi_lumaeq.h"
#include "lumaeq_kernels.hip.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "host/drain_guard.hpp"           // stand-alone helpers (CPU-testable: tests/cxx/test_host_helpers.cpp)
#include "host/copy_crew.hpp"
#include "host/pending_ranges.hpp"
#include "host/pin_registry.hpp"
#include "host/numa_affinity.hpp"
#include "host/wide_hint.hpp"

using namespace mi;

// The host side lives in host/*.inc.hpp, included in dependency order (ONE translation unit on purpose: the
// kernels are templates/inline device code and the anonymous-namespace helpers are shared).
#include "host/context.inc.hpp"          // opens the anonymous namespace of the launch helpers
#include "host/equalize_fused.inc.hpp"
#include "host/clahe.inc.hpp"
#include "host/host_forms.inc.hpp"       // closes it
#include "host/capi.inc.hpp"
#include "host/color.inc.hpp"
#include "host/clahe16.inc.hpp"
#include "host/pipe.inc.hpp"
#include "host/diff.inc.hpp"
