// rf_internal.h -- shared declarations of the recfilter_amd runtime (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/recfilter_amd.h"

// Developer A/B switches.  The shipped library reads NO environment variable: RF_KNOB("RF_...") is a null pointer unless the
// library is built with -DRF_AB_KNOBS (make AB=1 -> librecfilter_amd_ab.so, which the scripts under tools/ load through
// RECFILTER_AMD_LIB); what a caller may choose about a plan is rf_filter_desc.flags (include/recfilter_amd.h).
#ifdef RF_AB_KNOBS
#include <cstdlib>
#define RF_KNOB(name) getenv(name)
#else
#define RF_KNOB(name) ((const char *)nullptr)
#endif

namespace rf {

void set_error(const char *fmt, ...);

#define RF_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ::rf::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),      \
                            __FILE__, __LINE__);                                        \
            return RF_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

// ---------------------------------------------------------------------------------------
// Scan as the host sees it (coefficients kept in double for the table algebra).
struct Scan {
    int dim = 0;
    bool causal = true;
    int order = 0;                    // feedback taps of this scan
    double b = 0.0;                   // feedforward, already cast through the pixel type
    double a[RF_MAX_ORDER] = {0};     // feedback, already cast through the pixel type
    // Border modification (plan.cpp, "clamped sections"): mod_n < 0 -- the scan handles a clamped border natively (the
    // prologue of lib/recfilter.cpp:330-336); mod_n >= 0 -- the scan is run with a ZERO border, and on the tile where it
    // enters a clamped image its first mod_n samples x_r (in scan direction) are first replaced by x_r + mod_g[r] * x_0.
    int mod_n = -1;
    double mod_g[RF_MAX_ORDER] = {0};
};

// Coefficients as a kernel receives them, in the pixel's arithmetic type.
template <typename T>
struct ScanCoef {
    T b;
    T a[RF_MAX_ORDER];
};

// ---------------------------------------------------------------------------------------
// One launched kernel, for rf_plan_execute_timed / profilers.
struct KernelRecord {
    const char *name;
};

}  // namespace rf
