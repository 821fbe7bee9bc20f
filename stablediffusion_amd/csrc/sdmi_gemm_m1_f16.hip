// sdmi_gemm instantiation unit: mode 1 (implicit-GEMM conv3x3), f16 operands.
#include "sdmi_gemm_kernel.h"
SDMI_GEMM_UNIT(1, 1, m1_f16)
SDMI_GEMM_UNIT_SMALL_CONV(1, m1_f16)
