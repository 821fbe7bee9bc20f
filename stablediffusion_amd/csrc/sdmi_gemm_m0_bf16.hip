// sdmi_gemm instantiation unit: mode 0 (linear / conv1x1), bf16 operands.
#include "sdmi_gemm_kernel.h"
SDMI_GEMM_UNIT(0, 0, m0_bf16)
SDMI_GEMM_UNIT_SMALL(0, false, m0_bf16)
#if SDMI_DEBUG_CLOCKS
extern "C" int sdmi_debug_clocks_m0_bf16(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(sdmi_debug_clk), 2 * sizeof(long long)) == hipSuccess ? 0 : -1;
}
#endif
