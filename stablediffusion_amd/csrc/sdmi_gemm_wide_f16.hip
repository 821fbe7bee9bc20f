// sdmi_gemm instantiation unit: the 256 x 256 x 32 tile (sdmi_gemm_wide.h), f16 operands, both modes.
#include "sdmi_gemm_wide.h"
SDMI_GEMM_WIDE_UNIT(1, f16)
