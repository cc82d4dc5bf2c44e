// plan_fused.cpp -- plan for the LDS-staged fused x/y path (kernels_fused.hip).
#include "plan.h"

namespace rf {

bool fused_plan_applicable(const rf_plan *, const rf_filter_desc *, std::string *why) {
    if (why) *why = "not built yet";
    return false;
}

int build_fused_plan(rf_plan *, const rf_filter_desc *) {
    set_error("fused path not built yet");
    return RF_ERR_UNSUPPORTED;
}

}  // namespace rf
