// sdmi_gemm instantiation unit: mode 0 (linear / conv1x1), f16 operands.
#include "sdmi_gemm_kernel.h"
SDMI_GEMM_UNIT(0, 1, m0_f16)
SDMI_GEMM_UNIT_SMALL(1, false, m0_f16)
