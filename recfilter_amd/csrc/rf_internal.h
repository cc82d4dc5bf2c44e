// rf_internal.h -- shared declarations of the recfilter_amd runtime (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/recfilter_amd.h"

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
