// sdmi_gemm instantiation unit: mode 1 (implicit-GEMM conv3x3), bf16 operands.
#include "sdmi_gemm_kernel.h"
SDMI_GEMM_UNIT(1, 0, m1_bf16)
SDMI_GEMM_UNIT_SMALL_CONV(0, m1_bf16)
