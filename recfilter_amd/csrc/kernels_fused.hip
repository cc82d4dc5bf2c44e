// kernels_fused.hip -- placeholder, filled in next.
#include "kernels.h"
