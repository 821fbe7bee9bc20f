// sdmi_gemm instantiation unit: the LayerNorm-folded kernels (sdmi_gemm_args.ln_colsum; mode 0, f16 operands) - the tiled
// kernel's 128- / 256-row tiles and the 256 x 256 x 32 tile.
#include "sdmi_gemm_wide.h"
SDMI_GEMM_UNIT_LNF(m0_f16_lnf)
SDMI_GEMM_WIDE_UNIT_LNF(f16_lnf)
SDMI_GEMM_UNIT_SMALL(1, true, m0_f16_lnf)
