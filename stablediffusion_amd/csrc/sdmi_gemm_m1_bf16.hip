// sdmi_gemm instantiation unit: mode 1 (implicit-GEMM conv3x3), bf16 operands.
#include "sdmi_gemm_kernel.h"
SDMI_GEMM_UNIT(1, 0, m1_bf16)
SDMI_GEMM_UNIT_SMALL_CONV(0, m1_bf16)
#if SDMI_DEBUG_CLOCKS
extern "C" int sdmi_debug_clocks_m1_bf16(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sdmi_debug_clk), 2 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
